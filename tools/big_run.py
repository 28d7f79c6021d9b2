"""robustness probe: a scene several times C3 through the engine (not a benchmark)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("large-scale-sfm_amd.capi"); scene = importlib.import_module("large-scale-sfm_amd.scene")
m, n, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t0 = time.time(); sc = scene.make_scene("x", m=m, n=n, d=d, seed=5); t1 = time.time()
eng = capi.Engine(capi.default_config(itmax=1 << 30, disable_stops=1))
eng.set_timing_detail(2)
eng.load_scene(sc); eng.prepare(); t2 = time.time()
eng.begin(); r0 = np.sqrt(eng.info()[0] / sc.nobs)
for _ in range(3): eng.step()
eng.reset_timing(); t3 = time.time()
for _ in range(12): eng.step()
t4 = time.time()
tm = eng.timing(); info = eng.info()
print("m %d n %d obs %d: scene %.1fs plan+upload %.1fs; rmse %.3f -> %.4f; %.2f ms/step (build %.2f pcg %.2f (%.0f its) backsub %.2f)" %
      (m, n, sc.nobs, t1 - t0, t2 - t1, r0, np.sqrt(info[1] / sc.nobs), 1e3 * (t4 - t3) / 12, tm["build_ms"] / tm["build_launches"],
       tm["pcg_ms"] / tm["pcg_solves"], tm["pcg_iters"] / tm["pcg_solves"], tm["backsub_ms"] / tm["backsub_launches"]))
