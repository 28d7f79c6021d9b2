"""Numpy prototype (CPU, minutes): what a coarse space would buy the reduced solve.  Builds the dense reduced camera system of a ring
scene at its converged point for several dampings (through the CPU oracle -- test infrastructure; this is a probe, not product code) and
counts PCG iterations to sqrt(r'z / r0'z0) < 1e-3 with: block-Jacobi (what k_pcg_* runs), block-Jacobi + an additive coarse correction on
aggregates of g consecutive cameras x 7 parameters (plain and smoothed aggregation).  Results: profiles/r02_pcg_two_level_proto.txt.
    python tools/pcg_two_level_proto.py <cameras> <points> <views per point>"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
from oracle import Oracle
scene = importlib.import_module("large-scale-sfm_amd.scene")
import scipy.sparse as sp, scipy.sparse.linalg as spl, scipy.linalg as sl

def build(m, n, d, seed=5):
    sc = scene.make_scene("x", m=m, n=n, d=d, seed=seed)
    o = Oracle()
    r = o.run(sc, itmax=150, fd_jacobian=0)
    lin = o.linearize(sc, p=r["p"], fd_jacobian=0, want_S=False)
    return sc, lin, r

def reduced(sc, lin, mu):
    m, n, cnp = sc.m, sc.n, sc.cnp
    Vinv = np.linalg.inv(lin["V"] + mu * np.eye(3))
    pt = sc.obs_pt(); cam = sc.obs_cam
    W = lin["W"]                       # nobs x cnp x 3
    Y = np.einsum("oab,obc->oac", W, Vinv[pt])
    N = m * cnp
    S = np.zeros((m, m, cnp, cnp))
    E = lin["ea"].copy()
    np.add.at(E, cam, -np.einsum("oab,ob->oa", Y, lin["eb"][pt]))
    dmax = np.diff(sc.pt_ptr).max()
    # pairs within a point: all points have the same d here
    d = dmax
    obs = np.arange(sc.nobs).reshape(n, d)
    for a in range(d):
        for b in range(d):
            blk = np.einsum("nij,nkj->nik", Y[obs[:, a]], W[obs[:, b]])
            np.add.at(S, (cam[obs[:, a]], cam[obs[:, b]]), -blk)
    for j in range(m):
        S[j, j] += lin["U"][j] + mu * np.eye(cnp)
    return S.transpose(0, 2, 1, 3).reshape(N, N), E.reshape(-1)

def pcg(S, b, Minv, tol=1e-3, maxit=400):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz0 = rz = r @ z
    for it in range(1, maxit + 1):
        Sp = S @ p
        al = rz / (p @ Sp)
        x += al * p; r -= al * Sp
        z = Minv(r); rzn = r @ z
        if np.sqrt(abs(rzn) / rz0) < tol: return x, it
        p = z + (rzn / rz) * p; rz = rzn
    return x, maxit

GS = (16, 32, 64, 125)
if __name__ == "__main__":
    m, n, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    t = time.time(); sc, lin, r = build(m, n, d); print("solve", time.time() - t, "iters", r["ret"], "rmse", np.sqrt(r["info"][1] / sc.nobs))
    cnp = sc.cnp; N = m * cnp
    maxdiag = max(lin["U"].max(), lin["V"].max())
    for rel_mu in (1e-3, 1e-5, 1e-7, 1e-9):
        mu = rel_mu * maxdiag
        S, E = reduced(sc, lin, mu)
        # block Jacobi
        Dinv = np.stack([np.linalg.inv(S[j*cnp:(j+1)*cnp, j*cnp:(j+1)*cnp]) for j in range(m)])
        bj = lambda r_: np.einsum("jab,jb->ja", Dinv, r_.reshape(m, cnp)).reshape(-1)
        rhs = E if np.abs(E).max() > 0 else np.random.default_rng(0).normal(size=N)
        rhs = np.random.default_rng(0).normal(size=N)     # generic right-hand side: E is ~0 at the converged point
        _, it_bj = pcg(S, rhs, bj)
        out = ["mu %.0e bj %3d" % (rel_mu, it_bj)]
        for g in GS:
            ng = (m + g - 1) // g
            Z = np.zeros((N, ng * cnp))
            for j in range(m):
                Z[j*cnp:(j+1)*cnp, (j // g)*cnp:(j // g + 1)*cnp] = np.eye(cnp)
            Ac = Z.T @ S @ Z
            Acf = sl.cho_factor(Ac)
            add = lambda r_: bj(r_) + Z @ sl.cho_solve(Acf, Z.T @ r_)
            _, it_add = pcg(S, rhs, add)
            # smoothed aggregation: Zs = (I - w D^-1 S) Z
            DS = np.vstack([Dinv[j] @ S[j*cnp:(j+1)*cnp, :] for j in range(m)])
            Zs = Z - 0.66 * (DS @ Z)
            Acs = Zs.T @ S @ Zs; Acsf = sl.cho_factor(Acs)
            adds = lambda r_: bj(r_) + Zs @ sl.cho_solve(Acsf, Zs.T @ r_)
            _, it_sa = pcg(S, rhs, adds)
            out.append("g%d: add %3d sa %3d (nc %d)" % (g, it_add, it_sa, ng * cnp))
        print(" | ".join(out), flush=True)
