"""Golden vectors for the C++ plugin boundary from the REAL reference ParallelBA (CPU double build,
oracle/_ref/libpba_ref.so, `make -C oracle pba`): tests/golden/ref_pba.npz.

Every case is the seeded scene `helpers.pba_case(name)` pushed through tests/native/pba_plain_client.cpp -- the
same client source and header (include/mi_pba.h) the GPU test links against libmi_ba.so.  Stored per case: the
reference's successful iterations, final mean squared error, cameras (f, t, R, radial) and points, and a checksum
of the generated measurements (so a drifting generator is noticed on the GPU box).

Run here (needs /root/reference for the build only):  python tests/golden/make_golden_pba.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
import helpers  # noqa: E402


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "pba"])
    out = {}
    with tempfile.TemporaryDirectory() as wd:
        exe = helpers.build_pba_client(wd, "ref")
        for name in helpers.PBA_CASES:
            sc, kw = helpers.pba_case(name)
            prob = os.path.join(wd, name + ".txt")
            helpers.write_pba_problem(prob, sc, **kw)
            os.environ["PBA_CLIENT_ARGS"] = "-v 3" if name == "degenerate" else ""    # its normalisation factors are printed at this level
            r = helpers.run_pba_client(exe, prob, wd, -3, sc.m, sc.n)     # -3 = PBA_CPU_DOUBLE
            for line in r["stdout"].splitlines():
                if line.startswith("Depth normalized by"):
                    out[name + "_depth_scaling"] = np.float64(line.split()[3]); out[name + "_depth_median"] = np.float64(line.split()[4].strip("()"))
                if line.startswith("Focal length normalized by"):
                    out[name + "_focal_scaling"] = np.float64(line.split()[4])
            print(name, "iters", r["iters"], "mse", r["mse"], "f[:3]", r["f"][:3], "radial[2:5]", r["radial"][2:5])
            for k in ("iters", "mse", "f", "t", "R", "radial", "pts"):
                out[name + "_" + k] = np.asarray(r[k])
            out[name + "_xsum"] = np.float64(np.abs(sc.x).sum())
    path = os.path.join(HERE, "ref_pba.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
