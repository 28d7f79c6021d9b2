"""Similarity alignment of two point sets (include/mi_sim3.h; Merge3D3D.cpp:171-413): the CPU restatement against libc's
rand() and numpy, the HIP kernels against the restatement."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, pkg


class Sim3Oracle:
    def __init__(self):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        self.L = C.CDLL(os.path.join(ROOT, "oracle", "libba_oracle.so"))
        self.L.sim3_scale.restype = C.c_longlong

    def align(self, src, dst, rounds):
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        src = np.ascontiguousarray(src, np.float64); dst = np.ascontiguousarray(dst, np.float64)
        sc = C.c_double(); R = np.zeros(9); t = np.zeros(3); scores = np.zeros(rounds); bi = C.c_int()
        rc = self.L.sim3_align(len(src), P(src), P(dst), rounds, C.byref(sc), P(R), P(t), P(scores), C.byref(bi))
        return dict(rc=rc, scale=sc.value, R=R.reshape(3, 3), t=t, scores=scores, best_round=bi.value)


@pytest.fixture(scope="module")
def sim3_oracle():
    return Sim3Oracle()


def two_clusters(n=400, seed=0, outliers=25, scale=1.7, noise=0.004):
    """the same n points in two frames: dst = scale R src + t + noise, some gross mismatches, two coincident source points"""
    rng = np.random.default_rng(seed)
    src = rng.normal(size=(n, 3))
    w = np.array([0.3, -0.5, 0.8]); th = np.linalg.norm(w); k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    t = np.array([4.0, -2.0, 7.0])
    dst = scale * src @ R.T + t + rng.normal(scale=noise, size=(n, 3))
    dst[:outliers] += rng.normal(scale=3.0, size=(outliers, 3))
    src[n - 1] = src[n - 2]                                   # a pair with |src_i - src_j| = 0 is left out of the ratio list
    return src, dst, scale, R, t


def test_rand_restatement_is_libcs_rand(sim3_oracle):
    class S(C.Structure):
        _fields_ = [("ring", C.c_uint * 31), ("f", C.c_int), ("b", C.c_int)]
    libc = C.CDLL("libc.so.6")
    for seed in (10000, 10001, 14999, 1, 0, 987654321):
        st = S(); sim3_oracle.L.gl_srand(C.byref(st), seed); libc.srand(seed)
        assert [sim3_oracle.L.gl_rand(C.byref(st)) for _ in range(3000)] == [libc.rand() for _ in range(3000)]


def test_restatement_pieces_against_numpy(sim3_oracle):
    src, dst, scale, R, t = two_clusters()
    P = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    # scale: the middle element of the sorted ratio list
    sc = C.c_double()
    cnt = sim3_oracle.L.sim3_scale(len(src), P(src), P(dst), C.byref(sc))
    i, j = np.triu_indices(len(src), 1)
    ns = np.linalg.norm(src[i] - src[j], axis=1); nd = np.linalg.norm(dst[i] - dst[j], axis=1)
    ratios = np.sort(nd[ns != 0] / ns[ns != 0])
    assert cnt == len(ratios) == len(i) - 1 and abs(sc.value - ratios[len(ratios) // 2]) < 1e-14
    assert abs(sc.value - scale) < 0.01
    # polar factor
    rng = np.random.default_rng(1)
    for _ in range(20):
        H = rng.normal(size=(3, 3)) * rng.uniform(0.1, 100)
        Rp = np.zeros(9); sim3_oracle.L.sim3_polar(P(H.reshape(-1)), P(Rp))
        U, _, Vt = np.linalg.svd(H)
        assert np.abs(Rp.reshape(3, 3) - U @ Vt).max() < 1e-12
    # ComputeRT on a sample: median centres, no reflection fix
    idx = np.arange(0, 200, dtype=np.int32)
    Rr = np.zeros(9); tr = np.zeros(3)
    sim3_oracle.L.sim3_rt(len(idx), P(idx), P(src), P(dst), P(Rr), P(tr))
    cs = np.sort(src[idx], axis=0)[len(idx) // 2]; cd = np.sort(dst[idx], axis=0)[len(idx) // 2]
    U, _, Vt = np.linalg.svd((dst[idx] - cd).T @ (src[idx] - cs))
    assert np.abs(Rr.reshape(3, 3) - U @ Vt).max() < 1e-12 and np.abs(tr - (cd - U @ Vt @ cs)).max() < 1e-12
    # the sample of a round: size 10 doubling every 10 rounds, indices from srand(round + 10000)
    libc = C.CDLL("libc.so.6")
    buf = np.zeros(len(src), np.int32)
    for rnd, want in ((0, 10), (9, 10), (10, 20), (35, 80), (60, 400), (4999, 400)):
        assert sim3_oracle.L.sim3_sample(rnd, len(src), P(buf)) == want
    sim3_oracle.L.sim3_sample(0, len(src), P(buf))
    libc.srand(10000)
    first = libc.rand() % len(src)
    assert buf[0] == first


def test_alignment_recovers_the_transform(sim3_oracle):
    src, dst, scale, R, t = two_clusters(scale=1.0)            # with scale 1 the double scaling of the score is harmless
    r = sim3_oracle.align(src, dst, 300)
    assert r["rc"] == 0 and abs(r["scale"] - 1.0) < 0.01
    assert np.abs(r["R"] - R).max() < 0.02 and np.abs(r["t"] - t).max() < 0.05
    assert r["scores"][r["best_round"]] == r["scores"].min() and r["best_round"] == int(np.argmin(r["scores"]))


@pytest.mark.gpu
@pytest.mark.parametrize("n,rounds,scale", [(400, 300, 1.0), (1500, 120, 1.7), (37, 60, 0.6)])
def test_kernels_match_the_restatement(sim3_oracle, n, rounds, scale):
    src, dst, _, R, t = two_clusters(n=n, seed=n, outliers=n // 16, scale=scale)
    want = sim3_oracle.align(src, dst, rounds)
    got = pkg("capi").sim3_align(src, dst, rounds=rounds)
    assert got["scale"] == want["scale"]                       # the ratio list is the same list, the select is exact
    assert np.allclose(got["scores"], want["scores"], rtol=1e-10, atol=1e-13)
    b = got["best_round"]
    assert b == want["best_round"] or abs(want["scores"][b] - want["scores"].min()) < 1e-12 * want["scores"].min()
    if b == want["best_round"]:
        assert np.abs(got["R"] - want["R"]).max() < 1e-10 and np.abs(got["t"] - want["t"]).max() < 1e-9


@pytest.mark.gpu
def test_full_round_count_and_errors():
    capi = pkg("capi")
    src, dst, scale, R, t = two_clusters(n=3000, seed=5, outliers=150, scale=1.0)
    got = capi.sim3_align(src, dst)                            # 5000 rounds, 4.5 M ratios
    assert abs(got["scale"] - 1.0) < 0.01 and np.abs(got["R"] - R).max() < 0.01 and np.abs(got["t"] - t).max() < 0.05
    print("sim3: n = 3000, 5000 rounds, %.1f ms on the device" % got["ms"])
    # 12 000 points: 72 M ratios (576 MB) materialised and selected from, samples of up to 12 000 indices per round
    src, dst, scale, R, t = two_clusters(n=12000, seed=6, outliers=600, scale=1.0)
    got = capi.sim3_align(src, dst, rounds=400)
    assert abs(got["scale"] - 1.0) < 0.005 and np.abs(got["R"] - R).max() < 0.005 and np.abs(got["t"] - t).max() < 0.03
    print("sim3: n = 12000, 400 rounds, %.1f ms on the device" % got["ms"])
    with pytest.raises(RuntimeError, match="bad argument"):
        capi.sim3_align(src[:1], dst[:1])
    same = np.ones((5, 3))
    with pytest.raises(RuntimeError, match="no pair of distinct"):
        capi.sim3_align(same, dst[:5])
