"""probe: scenes whose points all see different camera sets (not a benchmark)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import conftest
capi = importlib.import_module("large-scale-sfm_amd.capi"); scene = importlib.import_module("large-scale-sfm_amd.scene")
kind, m, n, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
det = int(os.environ.get("MI_BA_DETERMINISTIC", "0"))
if kind == "ladybug":
    sc = scene.ladybug_like_scene(m=m, n=n, dmax=d, seed=4)
elif kind in scene.CONFIGS:
    sc = scene.make_scene(kind)
else:   # every point: d cameras drawn from a window of 3d neighbouring cameras
    base = scene.make_scene("x", m=m, n=n, d=3 * d, seed=6, strides=(1,))
    rng = np.random.default_rng(1)
    keep = np.sort(np.argsort(rng.random((n, 3 * d)), axis=1)[:, :d], axis=1) + (np.arange(n) * 3 * d)[:, None]
    keep = keep.reshape(-1)
    sc = scene.Scene(n=n, m=m, cnp=base.cnp, est_focal=1, undistort=0, explicit_centers=1, proj_sign=-1.0,
                     pt_ptr=(np.arange(n + 1) * d).astype(np.int32), obs_cam=base.obs_cam[keep], x=base.x[keep], R0=base.R0,
                     f_fixed=base.f_fixed, p0=base.p0, p_true=base.p_true)
    # cameras inside a point must ascend for the oracle-style CSR; re-sort per point
    oc = sc.obs_cam.reshape(n, d); order = np.argsort(oc, axis=1)
    sc.obs_cam = np.take_along_axis(oc, order, 1).reshape(-1).astype(np.int32)
    sc.x = np.take_along_axis(sc.x.reshape(n, d, 2), order[:, :, None], 1).reshape(-1, 2)
t1 = time.time()
eng = capi.Engine(capi.default_config(itmax=1 << 30, disable_stops=1, deterministic=det))
eng.set_timing_detail(2)
eng.load_scene(sc); eng.prepare(); t2 = time.time()
st = capi.Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam).stats()
eng.begin(); r0 = np.sqrt(eng.info()[0] / sc.nobs)
for _ in range(3): eng.step()
eng.reset_timing(); t3 = time.time()
for _ in range(10): eng.step()
t4 = time.time(); tm = eng.timing(); info = eng.info()
print("%s m %d n %d obs %d d_mean %.1f: plan %.1fs tiles %s batches %s; rmse %.3f -> %.4f; %.2f ms/step (build %.2f pcg %.2f (%.0f its) backsub %.2f)" %
      (kind, m, n, sc.nobs, sc.nobs / n, t2 - t1, st.get("ntile"), st.get("nbatch"), r0, np.sqrt(info[1] / sc.nobs), 1e3 * (t4 - t3) / 10,
       tm["build_ms"] / tm["build_launches"], tm["pcg_ms"] / tm["pcg_solves"], tm["pcg_iters"] / tm["pcg_solves"], tm["backsub_ms"] / tm["backsub_launches"]))
print({k: v for k, v in st.items()})
