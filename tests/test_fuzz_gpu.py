"""Randomised parity sweep: small scenes with ragged visibility (single views up to tracks longer than a wavefront, camera
sets that repeat and camera sets that never do -> register tiles, table tiles, row tiles and long points in one
problem), every parameterisation (cnp 6 / 7 / 8, both distortion models, t or C), bundle modes, fixed cameras, soft
constraints -- the engine against the CPU oracle (itself pinned to the reference), trial by trial."""
import os

import numpy as np
import pytest

from conftest import pkg

pytestmark = pytest.mark.gpu


def ragged_scene(rng, m, n, est_focal, undistort, explicit_centers, proj_sign, dmax):
    scene = pkg("scene")
    base = scene.make_scene("x", m=m, n=n, d=dmax, seed=int(rng.integers(1 << 30)), strides=(1,), est_focal=est_focal,
                            undistort=undistort, nk=1, explicit_centers=explicit_centers, proj_sign=proj_sign)
    keep = []
    for i in range(n):
        idx = np.arange(base.pt_ptr[i], base.pt_ptr[i + 1])
        r = rng.random()
        if r < 0.35:            # repeating camera sets: a prefix whose length depends on the start camera only
            d = 2 + int(base.obs_cam[idx[0]]) % 7
            sel = idx[:d]
        elif r < 0.85:          # never-repeating camera sets
            d = int(rng.integers(1, min(dmax, 18) + 1))
            sel = np.sort(rng.permutation(idx)[:d])
        else:                   # long tracks, some beyond one wavefront
            d = int(rng.integers(18, dmax + 1))
            sel = idx[:d]
        keep.append(sel)
    sel = np.concatenate(keep)
    cnt = np.array([len(k) for k in keep])
    return scene.Scene(n=n, m=m, cnp=base.cnp, est_focal=est_focal, undistort=undistort, nk=1, explicit_centers=explicit_centers,
                       proj_sign=proj_sign, pt_ptr=np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32),
                       obs_cam=base.obs_cam[sel], x=base.x[sel], R0=base.R0, f_fixed=base.f_fixed, p0=base.p0, p_true=base.p_true)


CASES = [  # est_focal, undistort, explicit_centers, proj_sign, mode, mcon, constraints
    (1, 0, 1, -1.0, 0, 0, False), (1, 0, 0, 1.0, 0, 2, False), (0, 0, 1, -1.0, 0, 0, False), (1, 1, 1, -1.0, 0, 0, False),
    (0, 1, 0, 1.0, 0, 1, False), (1, 2, 0, 1.0, 0, 2, False), (1, 0, 1, -1.0, 1, 0, False), (1, 0, 1, -1.0, 2, 0, False),
    (1, 0, 1, -1.0, 0, 0, True), (1, 1, 0, 1.0, 0, 0, True), (0, 2, 0, 1.0, 0, 1, False), (1, 0, 0, 1.0, 0, 3, True),
]


@pytest.mark.parametrize("det", [0, 1, 2])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_ragged_scenes_match_the_oracle(oracle, case, det, monkeypatch):
    """det = 1 (per-tile copies summed in tile order), 2 (every point on the row path): mi_ba_config.deterministic -- no f64 atomics, so the run is reproducible and the comparisons with the oracle take
    their strict form (every trial of the truncated drivers, every accept / reject decision, the tight parameter bound): what
    the atomics' run-to-run noise had forced to loosen (commits 0cd8ded, 0eba4de, 583573b)"""
    monkeypatch.setenv("MI_BA_MIN_RUN_PTS", "2")              # camera sets shared by two points already earn an LDS tile: all paths in one problem
    est_focal, undistort, explicit_centers, proj_sign, mode, mcon, cons = CASES[case]
    rng = np.random.default_rng(1000 + case + int(os.environ.get("MI_FUZZ_SEED_OFFSET", "0")))   # other scenes: set the offset
    sc = ragged_scene(rng, m=90, n=1200, est_focal=est_focal, undistort=undistort, explicit_centers=explicit_centers,
                      proj_sign=proj_sign, dmax=80)
    m, n, cnp = sc.m, sc.n, sc.cnp
    if cons:
        a0 = sc.p0[:m * cnp].reshape(m, cnp); X0 = sc.p0[m * cnp:].reshape(n, 3)
        cw = np.zeros((m, cnp)); cv = np.zeros((m, cnp))
        cw[::4, :3] = 30.0; cv[::4, :3] = a0[::4, :3] + 0.01
        if est_focal:
            cw[:, 6] = 2e4; cv[:, 6] = 0.8
        pw = np.zeros(n); pv = np.zeros((n, 3)); pw[::9] = 1e-3; pv[::9] = X0[::9] - 0.01
        sc.constraints = dict(cam_cw=cw, cam_cv=cv, pt_cw=pw, pt_cv=pv)
    capi = pkg("capi")
    modes = capi.Plan(n, m, sc.obs_pt(), sc.obs_cam).get("tile_mode")
    assert {0, 1, 2} <= set(modes.tolist())                   # every kind of tile takes part (det: the engine's own plan is all rows)
    eng = capi.Engine(capi.default_config(est_focal=est_focal, undistort=undistort, explicit_centers=explicit_centers,
                                          proj_sign=proj_sign, mode=mode, mcon=mcon, pcg_tol=1e-12, pcg_maxit=490, deterministic=det))
    eng.load_scene(sc)
    eng.prepare()
    ret, info = eng.run()
    o = oracle.run(sc, itmax=150, fd_jacobian=0, mcon=mcon, mode=mode)
    if mode == 0:                                             # (the truncated drivers end in a rounding-level tail whose
        assert ret == o["ret"] and info[6] == o["info"][6] and info[9] == o["info"][9]   # length the GPU's atomic order decides)
    assert abs(info[0] - o["info"][0]) < 1e-10 * o["info"][0]
    assert abs(info[1] - o["info"][1]) < 1e-7 * o["info"][1]
    tr, to = eng.trace(), o["trace"]
    # trial costs and accept / reject decisions; the truncated drivers finish with a few trials whose cost changes are at
    # rounding level, where the order of the GPU's atomic sums decides the sign of dF: those are left out
    if det:
        k = min(len(tr), len(to)) if mode else len(to)
        # every trial; the one decision left out is a trial whose cost equals the current cost to 11 digits -- its sign is the
        # CPU oracle's summation order against the GPU's, not noise: the GPU takes the same decision every run
        decided = np.abs(to[:k, 3] - to[:k, 2]) > 1e-11 * to[:k, 2]
        assert np.allclose(tr[:k, 3], to[:k, 3], rtol=1e-6) and np.array_equal(tr[:k, 4][decided], to[:k, 4][decided])
        if mcon or cons or mode:
            # (deterministic on the GPU, but not the CPU oracle's summation order: a camera with a handful of views ends ~1e-5 away
            #  along its weak directions -- 3e-5 seen with MI_FUZZ_SEED_OFFSET=600 -- at a cost equal to 1e-7)
            assert np.abs(eng.packed_params() - o["p"]).max() < (1e-3 if mode == 2 else 1e-4 if mode == 1 else 1e-5)
        return
    k = max(1, min(len(tr), len(to)) - 4) if mode else len(to)
    assert np.allclose(tr[:k, 3], to[:k, 3], rtol=1e-6)
    # (a trial that changes the cost by less than 1e-9 of it is accepted or rejected on rounding noise -- in the oracle too)
    decided = np.abs(to[:k, 3] - to[:k, 2]) > 1e-9 * to[:k, 2] if mode else np.ones(k, bool)
    assert np.array_equal(tr[:k, 4][decided], to[:k, 4][decided])
    if mcon or cons or mode:
        # (truncated drivers: a point seen once slides along its ray, a camera with a handful of views along its weak
        #  directions, at next to no cost -- where the rounding-level tail stops moves them by ~1e-4; the cost is the check)
        assert np.abs(eng.packed_params() - o["p"]).max() < (1e-3 if mode else 1e-5)


@pytest.mark.parametrize("m,n,d", [(2, 20, 2), (3, 9, 3), (9, 80, 2), (17, 33, 9)])
def test_tiny_problems(oracle, m, n, d):
    """the smallest problems sba accepts (2 cameras; barely more measurements than unknowns)"""
    sc = pkg("scene").make_scene("x", m=m, n=n, d=d, seed=m * 100 + n)
    capi = pkg("capi")
    eng = capi.Engine(capi.default_config(pcg_tol=1e-13, pcg_maxit=400))
    eng.load_scene(sc)
    eng.prepare()
    ret, info = eng.run()
    o = oracle.run(sc, itmax=150, fd_jacobian=0)
    assert abs(info[0] - o["info"][0]) < 1e-10 * o["info"][0]
    # few measurements per unknown: the normal equations are poorly conditioned and the tail of the run is rounding-level,
    # so compare where the cost ends up, not how many trials it took
    assert abs(info[1] - o["info"][1]) < 1e-6 * max(o["info"][1], 1e-12) or abs(np.sqrt(info[1] / sc.nobs) - np.sqrt(o["info"][1] / sc.nobs)) < 1e-6
    assert abs(oracle.cost(sc, eng.packed_params()) - info[1]) < 1e-9 * max(info[1], 1e-12)


@pytest.mark.parametrize("case", [0, 3, 5])
def test_pba_damping_on_ragged_scenes(oracle, case, monkeypatch):
    """mi_ba_config.lm_variant = 1 (PBA's damping on the diagonals, its gain ratio in the metric of diag J'J) where every kind of
    tile, the row path and tracks longer than a wavefront take part: reduced system, solution, back-substitution and the first
    trials of the controller against dense numpy on the oracle's blocks"""
    from test_engine_gpu import PBA_OPTS, reduced_system, rel
    monkeypatch.setenv("MI_BA_MIN_RUN_PTS", "2")
    est_focal, undistort, explicit_centers, proj_sign, mode, mcon, cons = CASES[case]
    rng = np.random.default_rng(5000 + case)
    sc = ragged_scene(rng, m=90, n=1200, est_focal=est_focal, undistort=undistort, explicit_centers=explicit_centers,
                      proj_sign=proj_sign, dmax=80)
    capi = pkg("capi")
    eng = capi.Engine(capi.default_config(est_focal=est_focal, undistort=undistort, explicit_centers=explicit_centers,
                                          proj_sign=proj_sign, pcg_tol=1e-13, pcg_maxit=490, pcg_minit=10, lm_variant=1, opts=PBA_OPTS, itmax=6))
    eng.load_scene(sc)
    eng.prepare()
    lin = oracle.linearize(sc, fd_jacobian=0)
    mu = 0.02
    g1 = eng.debug_linearize(mu)
    S, E, Vinv = reduced_system(lin, sc, mu, damp_diag=True)
    assert rel(g1["S"], S) < 1e-10 and rel(g1["E"].reshape(-1), E) < 1e-9
    g = eng.debug_solve(mu)
    da = np.linalg.solve(S, E).reshape(sc.m, sc.cnp)
    assert rel(g["da"], da) < 1e-6
    t = np.zeros((sc.n, 3))
    np.add.at(t, sc.obs_pt(), np.einsum("oab,oa->ob", lin["W"], da[sc.obs_cam]))
    db = np.einsum("iab,ib->ia", Vinv, lin["eb"] - t)
    assert rel(g["db"], db) < 1e-6
    # the controller's first trials: damping schedule and trial costs (the gain ratio's denominator is d'(mu D d + g))
    ret, info = eng.run()
    tr = eng.trace()
    p = sc.p0.copy(); mu, nu = PBA_OPTS[0], 2.0
    cost = oracle.cost(sc, p)
    for k in range(min(4, len(tr))):
        lin = oracle.linearize(sc, fd_jacobian=0, p=p)
        S, E, Vinv = reduced_system(lin, sc, mu, damp_diag=True)
        da = np.linalg.solve(S, E).reshape(sc.m, sc.cnp)
        t = np.zeros((sc.n, 3)); np.add.at(t, sc.obs_pt(), np.einsum("oab,oa->ob", lin["W"], da[sc.obs_cam]))
        db = np.einsum("iab,ib->ia", Vinv, lin["eb"] - t)
        DU = np.einsum("jaa->ja", lin["U"]); DV = np.einsum("iaa->ia", lin["V"])
        dL = (da * (mu * DU * da + lin["ea"])).sum() + (db * (mu * DV * db + lin["eb"])).sum()
        p_new = p + np.concatenate([da.reshape(-1), db.reshape(-1)])
        c_new = oracle.cost(sc, p_new)
        assert abs(tr[k, 1] - mu) < 1e-6 * mu and abs(tr[k, 3] - c_new) < 1e-7 * c_new
        assert bool(tr[k, 4]) == (c_new < cost)
        if c_new < cost:
            rho = (cost - c_new) / dL
            p, cost = p_new, c_new
            mu = max(mu * max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3), PBA_OPTS[3]); nu = 2.0
        else:
            mu *= nu; nu *= 2.0
