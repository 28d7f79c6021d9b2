"""ctypes binding of the C ABI in include/mi_ba.h (libmi_ba.so).

Plumbing only: every number comes from the HIP engine.  There is no CPU fallback -- a missing
library or a missing GPU raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmi_ba.so")
CSRC = os.path.join(HERE, "csrc")

INFOSZ = 10
PLAN_STATS = ["npts_local", "nobs_local", "nbatch", "ntile", "npair", "nb_upper", "nb_full", "nslot_total",
              "nbatch_rows", "npair_rows", "nlong", "pt_begin", "pt_end", "max_slots", "nrec", "nrow_pts"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Config(C.Structure):
    _fields_ = [("est_focal", C.c_int), ("undistort", C.c_int), ("explicit_centers", C.c_int), ("mcon", C.c_int),
                ("proj_sign", C.c_double), ("f_scale", C.c_double), ("k_scale", C.c_double),
                ("itmax", C.c_int), ("opts", C.c_double * 6), ("disable_stops", C.c_int),
                ("pcg_maxit", C.c_int), ("pcg_tol", C.c_double),
                ("device", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("verbose", C.c_int),
                ("tile_max_batches", C.c_int), ("mode", C.c_int), ("deterministic", C.c_int), ("nk", C.c_int), ("lm_variant", C.c_int), ("pcg_minit", C.c_int), ("reserved", C.c_int * 2)]


class Timing(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("build_ms", "build_launches", "pcg_ms", "pcg_solves", "pcg_iters", "backsub_ms", "backsub_launches",
                 "step_ms", "steps", "allreduce_ms", "allreduces", "build_bytes", "pcg_resumes", "build_flops", "pcg_timed", "spec_adopted", "build_index_bytes", "exchange_ms", "exchanges_timed", "replicated_ms", "replicated_timed", "pcg_persistent", "pcg_matrix_f32", "pcg_push", "build_block_handovers")] + [("reserved", C.c_double * 7)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


def build(force: bool = False) -> str:
    """compile libmi_ba.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    return LIB_PATH


_lib = None


HOST_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C {CSRC}); "
                               "the engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.mi_ba_create.restype = C.c_void_p
        L.mi_ba_last_error.restype = C.c_char_p
        L.mi_ba_last_error.argtypes = [C.c_void_p]
        L.mi_ba_plan_build.restype = C.c_void_p
        L.mi_ba_plan_build_rows.restype = C.c_void_p
        L.mi_ba_plan_get.restype = C.c_int64
        vp, dp, ip, dbl, i32, i64 = C.c_void_p, _dp, _ip, C.c_double, C.c_int, C.c_int64
        sig = {
            "mi_ba_destroy": [vp], "mi_ba_cnp": [vp], "mi_ba_prepare": [vp], "mi_ba_begin": [vp], "mi_ba_step": [vp], "mi_ba_restart": [vp],
            "mi_ba_reset_timing": [vp], "mi_ba_set_timing_detail": [vp, i32], "mi_ba_set_cameras": [vp, i32, dp, dp, dp], "mi_ba_set_points": [vp, i32, dp],
            "mi_ba_set_observations": [vp, i64, dp, ip, ip], "mi_ba_set_camera_constraints": [vp, dp, dp],
            "mi_ba_set_camera_links": [vp, C.c_int, ip, ip, dp],
            "mi_ba_comm_set_host_allreduce": [vp, HOST_ALLREDUCE, vp],
            "mi_ba_set_point_constraints": [vp, dp, dp], "mi_ba_set_fixed_cameras": [vp, C.c_void_p], "mi_ba_set_fixed_params": [vp, C.c_void_p], "mi_ba_run": [vp, dp], "mi_ba_get_info": [vp, dp],
            "mi_ba_get_trace": [vp, dp, i32], "mi_ba_get_params": [vp, dp, dp], "mi_ba_set_params": [vp, dp, dp],
            "mi_ba_get_rotations": [vp, dp], "mi_ba_debug_linearize": [vp, dbl, dp, dp, dp, dp, dp, dp, dp],
            "mi_ba_debug_solve": [vp, dbl, dp, dp, dp], "mi_ba_get_timing": [vp, C.POINTER(Timing)], "mi_ba_get_timing_n": [vp, C.POINTER(Timing), C.c_size_t],
            "mi_ba_comm_init": [vp, C.c_char_p], "mi_ba_comm_unique_id": [C.c_char_p], "mi_ba_comm_info": [vp, C.POINTER(C.c_int)],
            "mi_ba_plan_build": [i32, i32, i64, ip, ip, i32, i32, i32], "mi_ba_plan_build_rows": [i32, i32, i64, ip, ip, i32, i32, i32], "mi_ba_plan_free": [vp],
            "mi_ba_plan_stats": [vp, C.POINTER(i64)], "mi_ba_plan_get": [vp, C.c_char_p, vp, i64],
            "mi_ba_create": [C.POINTER(Config)], "mi_ba_default_config": [C.POINTER(Config)],
        }
        for name, args in sig.items():
            getattr(L, name).argtypes = args
        _lib = L
    return _lib


def default_config(**kw) -> Config:
    c = Config()
    lib().mi_ba_default_config(C.byref(c))
    for k, v in kw.items():
        if k == "opts":
            c.opts = (C.c_double * 6)(*v)
        else:
            setattr(c, k, v)
    return c


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Engine:
    """One engine = one GPU = one point shard."""

    def __init__(self, cfg: Config | None = None, **kw):
        self.L = lib()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self.h = self.L.mi_ba_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("mi_ba_create failed: " + self.L.mi_ba_last_error(None).decode())
        self.cnp = self.L.mi_ba_cnp(self.h)
        self.m = self.n = self.nobs = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.mi_ba_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, r, what):
        if r < 0:
            raise RuntimeError(f"{what}: " + self.L.mi_ba_last_error(self.h).decode())
        return r

    def error(self) -> str:
        return self.L.mi_ba_last_error(self.h).decode()

    def load_scene(self, sc, p0=None):
        """sc: large-scale-sfm_amd.scene.Scene (Bundler packing)."""
        p0 = sc.p0 if p0 is None else p0
        m, n, cnp = sc.m, sc.n, sc.cnp
        assert cnp == self.cnp, (cnp, self.cnp)
        self.set_problem(np.ascontiguousarray(p0[:m * cnp]).reshape(m, cnp), sc.R0, sc.f_fixed,
                         np.ascontiguousarray(p0[m * cnp:]).reshape(n, 3), sc.x, sc.obs_pt(), sc.obs_cam)
        cons = getattr(sc, "constraints", None) or {}
        self.set_constraints(cons.get("cam_cw"), cons.get("cam_cv"), cons.get("pt_cw"), cons.get("pt_cv"))

    def restart(self):
        self._chk(self.L.mi_ba_restart(self.h), "restart")

    def set_host_allreduce(self, fn):
        """test transport (mi_ba.h): fn(np.ndarray view, op) must reduce the buffer in place over all ranks (op 0 sum, 1 max)"""
        def thunk(_user, buf, count, op):
            try:
                fn(np.ctypeslib.as_array(buf, shape=(count,)), op)
                return 0
            except Exception:      # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        self._host_ar = HOST_ALLREDUCE(thunk)
        self._chk(self.L.mi_ba_comm_set_host_allreduce(self.h, self._host_ar, None), "set_host_allreduce")

    def set_camera_links(self, cam_a, cam_b, weights):
        """quadratic couplings w[q, r] (a[cam_a[q], r] - a[cam_b[q], r])^2 (PBA's shared-intrinsics constraints)"""
        a = np.ascontiguousarray(cam_a, np.int32); b = np.ascontiguousarray(cam_b, np.int32)
        w = np.ascontiguousarray(weights, np.float64)
        self._keep_links = (a, b, w)
        self._chk(self.L.mi_ba_set_camera_links(self.h, len(a), a.ctypes.data_as(_ip), b.ctypes.data_as(_ip), _d(w)), "set_camera_links")

    def set_fixed_params(self, mask):
        """mask[j, r] != 0: parameter r of camera j is held at its initial value (None clears)"""
        if mask is None:
            self._chk(self.L.mi_ba_set_fixed_params(self.h, None), "set_fixed_params")
            return
        mk = np.ascontiguousarray(mask, np.uint8)
        assert mk.size == self.m * self.cnp if hasattr(self, "m") else True
        self._chk(self.L.mi_ba_set_fixed_params(self.h, mk.ctypes.data_as(C.c_void_p)), "set_fixed_params")

    def set_constraints(self, cam_w=None, cam_v=None, pt_w=None, pt_v=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        cam_w, cam_v, pt_w, pt_v = c(cam_w), c(cam_v), c(pt_w), c(pt_v)
        self._chk(self.L.mi_ba_set_camera_constraints(self.h, _d(cam_w), _d(cam_v)), "set_camera_constraints")
        self._chk(self.L.mi_ba_set_point_constraints(self.h, _d(pt_w), _d(pt_v)), "set_point_constraints")

    def set_problem(self, a0, R0, f_fixed, xyz, xy, pt_idx, cam_idx):
        a0 = np.ascontiguousarray(a0, np.float64); R0 = np.ascontiguousarray(R0, np.float64)
        ff = None if f_fixed is None else np.ascontiguousarray(f_fixed, np.float64)
        xyz = np.ascontiguousarray(xyz, np.float64); xy = np.ascontiguousarray(xy, np.float64)
        pt = np.ascontiguousarray(pt_idx, np.int32); cam = np.ascontiguousarray(cam_idx, np.int32)
        self.m, self.n, self.nobs = a0.shape[0], xyz.shape[0], xy.shape[0]
        self._chk(self.L.mi_ba_set_cameras(self.h, self.m, _d(a0), _d(R0), _d(ff)), "set_cameras")
        self._chk(self.L.mi_ba_set_points(self.h, self.n, _d(xyz)), "set_points")
        self._chk(self.L.mi_ba_set_observations(self.h, self.nobs, _d(xy), pt.ctypes.data_as(_ip),
                                                cam.ctypes.data_as(_ip)), "set_observations")

    def comm_info(self) -> dict:
        """what RCCL reports about this rank's communicator (mi_ba_comm_info)"""
        out = (C.c_int * 4)()
        self._chk(self.L.mi_ba_comm_info(self.h, out), "comm_info")
        return {"nranks": int(out[0]), "device": int(out[1]), "rank": int(out[2]), "rccl": bool(out[3])}

    def comm_init(self, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        self._chk(self.L.mi_ba_comm_init(self.h, buf), "comm_init")

    def prepare(self):
        self._chk(self.L.mi_ba_prepare(self.h), "prepare")

    def run(self):
        info = np.zeros(INFOSZ)
        ret = self.L.mi_ba_run(self.h, _d(info))
        return ret, info

    def begin(self):
        self._chk(self.L.mi_ba_begin(self.h), "begin")

    def step(self) -> int:
        return self._chk(self.L.mi_ba_step(self.h), "step")

    def info(self):
        info = np.zeros(INFOSZ)
        self.L.mi_ba_get_info(self.h, _d(info))
        return info

    def trace(self):
        n = self.L.mi_ba_get_trace(self.h, None, 0)
        t = np.zeros((max(n, 1), 6))
        self.L.mi_ba_get_trace(self.h, _d(t), n)
        return t[:n]

    def params(self):
        a = np.zeros((self.m, self.cnp)); xyz = np.zeros((self.n, 3))
        self._chk(self.L.mi_ba_get_params(self.h, _d(a), _d(xyz)), "get_params")
        return a, xyz

    def packed_params(self):
        a, xyz = self.params()
        return np.concatenate([a.reshape(-1), xyz.reshape(-1)])

    def set_params(self, a, xyz):
        a = np.ascontiguousarray(a, np.float64); xyz = np.ascontiguousarray(xyz, np.float64)
        self._chk(self.L.mi_ba_set_params(self.h, _d(a), _d(xyz)), "set_params")

    def rotations(self):
        R = np.zeros((self.m, 9))
        self._chk(self.L.mi_ba_get_rotations(self.h, _d(R)), "get_rotations")
        return R

    def debug_linearize(self, mu):
        N = self.m * self.cnp
        S = np.zeros((N, N)); E = np.zeros((self.m, self.cnp)); ea = np.zeros((self.m, self.cnp))
        eb = np.zeros((self.n, 3)); V = np.zeros((self.n, 3, 3)); dU = np.zeros((self.m, self.cnp)); cost = C.c_double(0)
        self._chk(self.L.mi_ba_debug_linearize(self.h, mu, _d(S), _d(E), _d(ea), _d(eb), _d(V), _d(dU),
                                               C.cast(C.byref(cost), _dp)), "debug_linearize")
        return dict(S=S, E=E, ea=ea, eb=eb, V=V, diagU=dU, cost=cost.value)

    def debug_solve(self, mu):
        da = np.zeros((self.m, self.cnp)); db = np.zeros((self.n, 3)); cost = C.c_double(0)
        its = self._chk(self.L.mi_ba_debug_solve(self.h, mu, _d(da), _d(db), C.cast(C.byref(cost), _dp)), "debug_solve")
        return dict(da=da, db=db, cost_new=cost.value, pcg_iters=its)

    def timing(self) -> dict:
        t = Timing()
        self.L.mi_ba_get_timing_n(self.h, C.byref(t), C.sizeof(Timing))
        return t.as_dict()

    def reset_timing(self):
        self.L.mi_ba_reset_timing(self.h)

    def set_timing_detail(self, level: int):
        """0 (default) no HIP events, 1 around the build kernels, 2 also around the reduced solve and the back-substitution"""
        self.L.mi_ba_set_timing_detail(self.h, int(level))


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    if lib().mi_ba_comm_unique_id(buf) != 0:
        raise RuntimeError("mi_ba_comm_unique_id failed")
    return buf.raw


class Plan:
    """host-only shard plan (usable without a GPU)."""

    _dtypes = dict(obs_src=np.int64, batch_pair=np.int64, pair=np.uint32, rpair=np.uint32, wk_q=np.int64, sb_q=np.int64, sb8_q=np.int64, obs_cslot=np.uint8,
                   tile_mode=np.uint8, bsr_trans=np.uint8)

    def __init__(self, n, m, pt_idx, cam_idx, rank=0, world=1, tile_max_batches=0, all_rows=False):
        self.L = lib()
        pt = np.ascontiguousarray(pt_idx, np.int32); cam = np.ascontiguousarray(cam_idx, np.int32)
        build = self.L.mi_ba_plan_build_rows if all_rows else self.L.mi_ba_plan_build
        self.h = build(n, m, len(pt), pt.ctypes.data_as(_ip), cam.ctypes.data_as(_ip), rank, world, tile_max_batches)
        if not self.h:
            raise ValueError("plan: " + self.L.mi_ba_last_error(None).decode())

    def stats(self) -> dict:
        s = (C.c_int64 * 16)()
        self.L.mi_ba_plan_stats(self.h, s)
        return {k: int(s[i]) for i, k in enumerate(PLAN_STATS)}

    def get(self, name: str) -> np.ndarray:
        n = self.L.mi_ba_plan_get(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(max(n, 1), self._dtypes.get(name, np.int32))
        self.L.mi_ba_plan_get(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), n)
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mi_ba_plan_free(self.h)
            self.h = None


# ---- multi-view triangulation (include/mi_triangulate.h) -------------------------------------------------------
class TriConfig(C.Structure):
    _fields_ = [("min_views", C.c_int), ("min_angle_deg", C.c_double), ("max_error", C.c_double),
                ("float_normalise", C.c_int), ("polish", C.c_int), ("device", C.c_int), ("reserved", C.c_int * 6)]


class Triangulator:
    """tracks -> 3-D points on the GPU, the pipeline's TriangulationMultiStep loop"""

    def __init__(self, **kw):
        L = self.L = lib()
        L.mi_tri_create.restype = C.c_void_p; L.mi_tri_create.argtypes = [C.POINTER(TriConfig)]
        L.mi_tri_last_error.restype = C.c_char_p; L.mi_tri_last_error.argtypes = [C.c_void_p]
        L.mi_tri_destroy.argtypes = [C.c_void_p]; L.mi_tri_run.argtypes = [C.c_void_p]
        L.mi_tri_set_cameras.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp]
        L.mi_tri_set_tracks.argtypes = [C.c_void_p, C.c_int, _ip, _ip, _dp]
        L.mi_tri_get.argtypes = [C.c_void_p, _dp, _dp, _ip, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.mi_tri_last_ms.restype = C.c_double; L.mi_tri_last_ms.argtypes = [C.c_void_p]
        L.mi_tri_algorithmic_bytes.restype = C.c_double; L.mi_tri_algorithmic_bytes.argtypes = [C.c_void_p]
        cfg = TriConfig()
        L.mi_tri_default_config(C.byref(cfg))
        for k, v in kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = L.mi_tri_create(C.byref(cfg))
        if not self.h:
            raise RuntimeError("mi_tri_create failed: " + L.mi_tri_last_error(None).decode())

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(what + ": " + self.L.mi_tri_last_error(self.h).decode())

    def set_cameras(self, f, R, t, c=None):
        f = np.ascontiguousarray(f, np.float64); R = np.ascontiguousarray(R, np.float64).reshape(-1)
        t = np.ascontiguousarray(t, np.float64).reshape(-1)
        c = None if c is None else np.ascontiguousarray(c, np.float64).reshape(-1)
        self._chk(self.L.mi_tri_set_cameras(self.h, len(f), _d(f), _d(R), _d(t), _d(c)), "set_cameras")

    def set_tracks(self, pt_ptr, obs_cam, xy):
        self._keep = (np.ascontiguousarray(pt_ptr, np.int32), np.ascontiguousarray(obs_cam, np.int32),
                      np.ascontiguousarray(xy, np.float64).reshape(-1))
        p, o, x = self._keep
        self.n = len(p) - 1
        self._chk(self.L.mi_tri_set_tracks(self.h, self.n, p.ctypes.data_as(_ip), o.ctypes.data_as(_ip), _d(x)), "set_tracks")

    def run(self):
        self._chk(self.L.mi_tri_run(self.h), "run")
        return self.L.mi_tri_last_ms(self.h)

    def get(self):
        n = self.n
        xyz = np.zeros((n, 3)); err = np.zeros(n); status = np.zeros(n, np.int32)
        counts = (C.c_int64 * 4)(); s = C.c_double(0); npj = C.c_int64(0)
        self._chk(self.L.mi_tri_get(self.h, _d(xyz), _d(err), status.ctypes.data_as(_ip), counts, C.byref(s), C.byref(npj)), "get")
        return dict(xyz=xyz, err=err, status=status, counts=list(counts), sum_sq_px=s.value, nproj=npj.value)

    def algorithmic_bytes(self):
        return self.L.mi_tri_algorithmic_bytes(self.h)

    def close(self):
        if self.h:
            self.L.mi_tri_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- similarity alignment of two point sets (include/mi_sim3.h) -------------------------------------------------
class Sim3Config(C.Structure):
    _fields_ = [("rounds", C.c_int), ("device", C.c_int), ("reserved", C.c_int * 6)]


def sim3_align(src, dst, rounds=5000, device=0):
    """ComputeRigidTransform of the cluster-merge step on the GPU -> dict(scale, R, t, scores, best_round, ms)"""
    L = lib()
    L.mi_sim3_last_error.restype = C.c_char_p
    L.mi_sim3_align.argtypes = [C.POINTER(Sim3Config), C.c_int, _dp, _dp, C.POINTER(C.c_double), _dp, _dp, _dp,
                                C.POINTER(C.c_int), C.POINTER(C.c_double)]
    src = np.ascontiguousarray(src, np.float64); dst = np.ascontiguousarray(dst, np.float64)
    cfg = Sim3Config()
    L.mi_sim3_default_config(C.byref(cfg))
    cfg.rounds = rounds; cfg.device = device
    scale = C.c_double(); R = np.zeros(9); t = np.zeros(3); scores = np.zeros(rounds); best = C.c_int(); ms = C.c_double()
    rc = L.mi_sim3_align(C.byref(cfg), len(src), _d(src.reshape(-1)), _d(dst.reshape(-1)), C.byref(scale), _d(R), _d(t), _d(scores),
                         C.byref(best), C.byref(ms))
    if rc != 0:
        raise RuntimeError("mi_sim3_align: " + L.mi_sim3_last_error().decode())
    return dict(scale=scale.value, R=R.reshape(3, 3), t=t, scores=scores, best_round=best.value, ms=ms.value)
