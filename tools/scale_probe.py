"""probe: a scene k times the size of C3 (cameras, points and observations all x k) -- how the trial's parts scale (not a benchmark)
   python tools/scale_probe.py <k>"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("large-scale-sfm_amd.capi"); scene = importlib.import_module("large-scale-sfm_amd.scene")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t = time.time()
sc = scene.make_scene("x", m=1000 * k, n=1000000 * k, d=10, seed=20262)
print("scene %.1f s: m %d n %d obs %d" % (time.time() - t, sc.m, sc.n, sc.nobs), flush=True)
eng = capi.Engine(capi.default_config(itmax=1 << 30, disable_stops=1))
eng.set_timing_detail(2)
t = time.time(); eng.load_scene(sc); eng.prepare(); print("load + prepare %.2f s" % (time.time() - t), flush=True)
eng.begin(); r0 = np.sqrt(eng.info()[0] / sc.nobs)
for _ in range(5): eng.step()
eng.reset_timing(); t = time.time()
for _ in range(20): eng.step()
dt = time.time() - t; tm = eng.timing(); info = eng.info()
print("rmse %.3f -> %.4f; %.2f ms/step (build %.2f pcg %.2f (%.1f its) backsub %.2f)" %
      (r0, np.sqrt(info[1] / sc.nobs), 1e3 * dt / 20, tm["build_ms"] / tm["build_launches"], tm["pcg_ms"] / max(tm["pcg_timed"], 1),
       tm["pcg_iters"] / tm["pcg_solves"], tm["backsub_ms"] / tm["backsub_launches"]))
