"""Golden vectors for the triangulation step: tests/golden/ref_tri.npz.

For every track of the seeded scene `helpers.tri_scene()` with at least two usable views:
  * X_ref / err_ref : what the REFERENCE's own triangulate_n returns (Bundler's imagelib/triangulate.c compiled, with
    the reference's matrix.c, cminpack and MKL's dgelsy, into oracle/_ref/libref_sfm.so by oracle/Makefile) -- on this
    platform the linear solution, because the reference's lmdif polish is a no-op here (oracle/tri_oracle.c);
  * X_minpack       : the real MINPACK lmdif (scipy.optimize.leastsq with lmdif1's settings: ftol = xtol = 1e-5,
    gtol 0, factor 100, maxfev 800) started from X_ref -- the pin of the `polish = 1` restatement.
The inputs are regenerated from the seed by the tests; a checksum of the measurements is stored.
"""
import ctypes as C
import os
import sys

import numpy as np
from scipy.optimize import leastsq

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
import helpers  # noqa: E402


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_sfm.so"))
    ts = helpers.tri_scene()
    n = ts["n"]
    R = ts["R"].reshape(-1, 3, 3)
    X_ref = np.zeros((n, 3)); err_ref = np.full(n, -1.0); X_mp = np.zeros((n, 3))
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    for i in range(n):
        o = np.arange(ts["pt_ptr"][i], ts["pt_ptr"][i + 1])
        o = o[ts["obs_cam"][o] >= 0]
        if len(o) < 2:
            continue
        cams = ts["obs_cam"][o]
        pv = np.ascontiguousarray((ts["xy"][o] / ts["f"][cams, None]).astype(np.float32).astype(np.float64))
        Rs = np.ascontiguousarray(R[cams].reshape(-1)); tt = np.ascontiguousarray(ts["t"][cams].reshape(-1))
        X = np.zeros(3); e = np.zeros(1)
        ref.ref_triangulate_n(len(o), dp(pv), dp(Rs), dp(tt), dp(X), dp(e))
        X_ref[i] = X; err_ref[i] = e[0]

        def res(Y):
            q = np.einsum("kij,j->ki", R[cams], Y) + ts["t"][cams]
            return (pv - q[:, :2] / q[:, 2:3]).reshape(-1)
        X_mp[i] = leastsq(res, X.copy(), ftol=1e-5, xtol=1e-5, gtol=0.0, maxfev=800, factor=100)[0]
    path = os.path.join(HERE, "ref_tri.npz")
    np.savez_compressed(path, X_ref=X_ref, err_ref=err_ref, X_minpack=X_mp, xysum=np.abs(ts["xy"]).sum())
    print(path, os.path.getsize(path) // 1024, "KiB; tracks", n, "triangulated", int((err_ref >= 0).sum()),
          "max |minpack - linear|", np.abs(X_mp - X_ref).max())


if __name__ == "__main__":
    main()
