"""ctypes bindings for the CPU checker (TEST INFRASTRUCTURE, see oracle/ba_oracle.h).

``Oracle``  -- the in-repo C restatement (oracle/libba_oracle.so, built by oracle/Makefile).
``Reference`` -- the real reference solver (oracle/_ref/libref_sfm.so), present only where it
was built from /root/reference; ``Reference.available()`` says whether it can be loaded.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from contextlib import contextmanager

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libba_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libref_sfm.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class _Problem(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("mcon", C.c_int), ("cnp", C.c_int),
                ("est_focal", C.c_int), ("undistort", C.c_int), ("explicit_centers", C.c_int),
                ("proj_sign", C.c_double), ("f_scale", C.c_double), ("k_scale", C.c_double),
                ("pt_ptr", _ip), ("obs_cam", _ip), ("x", _dp), ("R0", _dp), ("f_fixed", _dp),
                ("cam_cw", _dp), ("cam_cv", _dp), ("pt_cw", _dp), ("pt_cv", _dp), ("nk", C.c_int)]


class _Options(C.Structure):
    _fields_ = [("itmax", C.c_int), ("fd_jacobian", C.c_int), ("solver", C.c_int),
                ("pcg_maxit", C.c_int), ("pcg_tol", C.c_double), ("opts", C.c_double * 6),
                ("disable_stops", C.c_int), ("mode", C.c_int)]


#: Bundler's options, sfm.c:711-719 (eps2 is run_sfm's argument; Bundler passes 1e-12)
BUNDLER_OPTS = (1.0e-3, 1.0e-10, 1.0e-12, 1.0e-12, 0.0, 4.0e-2)
TRACE_W = 6


def build(force: bool = False) -> None:
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "ba_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


@contextmanager
def quiet_stdout(to: str | None = None):
    """silence printf from C (the reference prints every accepted step, sba_levmar.c:1563), or keep it in a file."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY) if to is None else os.open(to, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.ora_motstr_levmar.restype = C.c_int
        self.lib.ora_linearize.restype = C.c_double
        self.lib.ora_residuals.restype = C.c_double

    @staticmethod
    def _problem(sc, mcon=0):
        keep = dict(pt_ptr=np.ascontiguousarray(sc.pt_ptr, np.int32), obs_cam=np.ascontiguousarray(sc.obs_cam, np.int32),
                    x=np.ascontiguousarray(sc.x, np.float64), R0=np.ascontiguousarray(sc.R0, np.float64),
                    f_fixed=np.ascontiguousarray(sc.f_fixed, np.float64))
        for k in ("cam_cw", "cam_cv", "pt_cw", "pt_cv"):     # optional soft constraints (Scene.constraints dict)
            v = (getattr(sc, "constraints", None) or {}).get(k)
            keep[k] = None if v is None else np.ascontiguousarray(v, np.float64)
        pb = _Problem(sc.n, sc.m, mcon, sc.cnp, sc.est_focal, sc.undistort, sc.explicit_centers,
                      sc.proj_sign, 0.001, 5.0, _i(keep["pt_ptr"]), _i(keep["obs_cam"]),
                      _d(keep["x"]), _d(keep["R0"]), _d(keep["f_fixed"]),
                      _d(keep["cam_cw"]), _d(keep["cam_cv"]), _d(keep["pt_cw"]), _d(keep["pt_cv"]), getattr(sc, "nk", 2))
        return pb, keep

    def run(self, sc, p0=None, itmax=150, fd_jacobian=1, solver=0, pcg_maxit=200, pcg_tol=1e-10,
            opts=BUNDLER_OPTS, mcon=0, disable_stops=0, mode=0):
        pb, keep = self._problem(sc, mcon)
        op = _Options(itmax, fd_jacobian, solver, pcg_maxit, pcg_tol, (C.c_double * 6)(*opts), disable_stops, mode)
        p = np.array(sc.p0 if p0 is None else p0, np.float64, copy=True)
        info = np.zeros(10)
        cap = 16 * max(itmax, 1) + 16
        trace = np.zeros((cap, TRACE_W))
        nt = C.c_int(0)
        ret = self.lib.ora_motstr_levmar(C.byref(pb), C.byref(op), _d(p), _d(info), _d(trace), cap, C.byref(nt))
        return dict(ret=ret, p=p, info=info, trace=trace[:nt.value].copy())

    def linearize(self, sc, p=None, fd_jacobian=0, mcon=0, want_S=True):
        pb, keep = self._problem(sc, mcon)
        p = np.ascontiguousarray(sc.p0 if p is None else p, np.float64)
        m, n, cnp, nobs = sc.m, sc.n, sc.cnp, sc.nobs
        U = np.zeros((m, cnp, cnp)); V = np.zeros((n, 3, 3)); W = np.zeros((nobs, cnp, 3))
        S = np.zeros(((m - mcon) * cnp, (m - mcon) * cnp)) if want_S else None
        ea = np.zeros((m, cnp)); eb = np.zeros((n, 3)); e = np.zeros((nobs, 2)); jac = np.zeros((nobs, 2 * cnp + 6))
        cost = self.lib.ora_linearize(C.byref(pb), fd_jacobian, _d(p), _d(U), _d(V), _d(W), _d(S), _d(ea), _d(eb), _d(e), _d(jac))
        return dict(cost=cost, U=U, V=V, W=W, S=S, ea=ea, eb=eb, e=e, jac=jac)

    def cost(self, sc, p):
        pb, keep = self._problem(sc)
        p = np.ascontiguousarray(p, np.float64)
        return self.lib.ora_residuals(C.byref(pb), _d(p), None)


class Reference:
    """The real sba-1.5 + Bundler driver (reference code, compiled by oracle/Makefile `ref`)."""

    @staticmethod
    def available() -> bool:
        if not os.path.exists(REF_SO):
            return False
        try:
            C.CDLL(REF_SO)
            return True
        except OSError:
            return False

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        self.lib.ref_sba_motstr.restype = C.c_int

    def run(self, sc, p0=None, itmax=150, opts=BUNDLER_OPTS, mcon=0, verbose=0, want_blocks=False, log_path=None):
        """log_path: file that receives the reference's own printf output (verbose 2: one line per damping trial)"""
        p = np.array(sc.p0 if p0 is None else p0, np.float64, copy=True)
        pt_ptr = np.ascontiguousarray(sc.pt_ptr, np.int32); obs_cam = np.ascontiguousarray(sc.obs_cam, np.int32)
        x = np.ascontiguousarray(sc.x, np.float64); R0 = np.ascontiguousarray(sc.R0, np.float64)
        ff = np.ascontiguousarray(sc.f_fixed, np.float64)
        info = np.zeros(10)
        o6 = (C.c_double * 6)(*opts)
        m, n, cnp = sc.m, sc.n, sc.cnp
        V = S = U = W = None
        if want_blocks:
            V = np.zeros((n, 3, 3)); S = np.zeros((m * cnp, m * cnp)); U = np.zeros((m, cnp, cnp))
            W = np.zeros((m * cnp, n * 3))
        with quiet_stdout(log_path):
            ret = self.lib.ref_sba_motstr(n, m, mcon, cnp, _i(pt_ptr), _i(obs_cam), _d(x), _d(p), _d(R0), _d(ff),
                                          sc.est_focal, sc.undistort, sc.explicit_centers, itmax, verbose, o6,
                                          _d(info), _d(V), _d(S), _d(U), _d(W), getattr(sc, "nk", 2),
                                          *[_d(None if (getattr(sc, "constraints", None) or {}).get(k) is None
                                               else np.ascontiguousarray(sc.constraints[k], np.float64))
                                            for k in ("cam_cw", "cam_cv", "pt_cw", "pt_cv")])
        return dict(ret=ret, p=p, info=info, V=V, S=S, U=U, W=W)

    def run_partial(self, sc, mode, itmax=150, opts=BUNDLER_OPTS, mcon=0):
        """mode 1: sba_mot_levmar (points fixed), mode 2: sba_str_levmar (cameras fixed); returns the full p"""
        m, n, cnp = sc.m, sc.n, sc.cnp
        cams = np.ascontiguousarray(sc.p0[:m * cnp]).copy(); pts = np.ascontiguousarray(sc.p0[m * cnp:]).copy()
        pt_ptr = np.ascontiguousarray(sc.pt_ptr, np.int32); obs_cam = np.ascontiguousarray(sc.obs_cam, np.int32)
        x = np.ascontiguousarray(sc.x, np.float64); R0 = np.ascontiguousarray(sc.R0, np.float64)
        ff = np.ascontiguousarray(sc.f_fixed, np.float64)
        info = np.zeros(10)
        free, fixed = (cams, pts) if mode == 1 else (pts, cams)
        self.lib.ref_sba_partial.restype = C.c_int
        with quiet_stdout():
            ret = self.lib.ref_sba_partial(mode, n, m, mcon, cnp, _i(pt_ptr), _i(obs_cam), _d(x), _d(free), _d(fixed), _d(R0), _d(ff),
                                           sc.est_focal, sc.explicit_centers, itmax, (C.c_double * 6)(*opts), _d(info))
        return dict(ret=ret, p=np.concatenate([cams, pts]), info=info)

    def run_sfm(self, sc, eps2=1e-12, cam_w=None, cam_v=None, pt_v=None, pt_weight=0.0):
        """Bundler's run_sfm end to end (MAX_ITERS 150 hard-coded, sfm.c:820); cam_w / cam_v (m, 9) and pt_v (n, 3)
        are run_sfm's own soft constraints in camera_params_t units."""
        m, n, cnp = sc.m, sc.n, sc.cnp
        a = sc.p0[:m * cnp].reshape(m, cnp)
        R = np.ascontiguousarray(sc.R0, np.float64).copy()
        t = np.ascontiguousarray(a[:, :3]).copy()
        f = (a[:, 6] / 0.001).copy() if sc.est_focal else np.ascontiguousarray(sc.f_fixed).copy()
        k = np.zeros((m, 2))
        pts = np.ascontiguousarray(sc.p0[m * cnp:]).copy()
        pt_ptr = np.ascontiguousarray(sc.pt_ptr, np.int32); obs_cam = np.ascontiguousarray(sc.obs_cam, np.int32)
        x = np.ascontiguousarray(sc.x, np.float64)
        with quiet_stdout():
            opt = [None if v is None else np.ascontiguousarray(v, np.float64) for v in (cam_w, cam_v, pt_v)]
            self.lib.ref_run_sfm_cons(n, m, 0, _i(pt_ptr), _i(obs_cam), _d(x), sc.est_focal, sc.undistort,
                                      sc.explicit_centers, _d(R), _d(t), _d(f), _d(k), _d(pts), C.c_double(eps2),
                                      *[None if v is None else _d(v) for v in opt], C.c_double(pt_weight))
        return dict(R=R.reshape(m, 3, 3), t=t, f=f, k=k, pts=pts.reshape(n, 3))
