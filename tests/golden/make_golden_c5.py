"""tests/golden/ref_C5.npz: the REAL reference (oracle/_ref/libref_sfm.so: sba-1.5 driven like Bundler's run_sfm) on BASELINE config C5 as
tests/test_boundary_gpu.py::test_config_c5_merged_clusters_global_refine runs it -- the 160-camera scene merged from 8 sub-reconstructions
with residual Sim(3) seams (tests/helpers.py c5_scene), the iteration cap optimizer_all is called with there (100).  The reference solves
the problem in Bundler's -z frame (helpers.to_bundler_frame: a rotation of the camera frames, the same trajectory up to rounding).
Run in the build container only (make -C oracle ref first):   python tests/golden/make_golden_c5.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import Reference  # noqa: E402
from helpers import c5_scene, to_bundler_frame  # noqa: E402

sc, p_init = c5_scene()
sb, pb = to_bundler_frame(sc, p_init)
ref = Reference()
full = ref.run(sb, p0=pb, itmax=100)
costs = np.array([ref.run(sb, p0=pb, itmax=k)["info"][1] for k in range(1, int(full["ret"]) + 1)])
m, cnp = sb.m, sb.cnp
path = os.path.join(HERE, "ref_C5.npz")
np.savez_compressed(path, xsum=np.abs(sb.x).sum(), psum=np.abs(pb).sum(), info_final=full["info"], ret_final=full["ret"], cost_trace=costs,
                    cams_final=full["p"][:m * cnp], pts_final_head=full["p"][m * cnp:m * cnp + 6000])
print("C5 ->", path, os.path.getsize(path) // 1024, "KiB;", sb.nobs, "observations; iterations", full["ret"], "stop", full["info"][6], "nlss", full["info"][9],
      "rmse", np.sqrt(full["info"][0] / sb.nobs), "->", np.sqrt(full["info"][1] / sb.nobs))
