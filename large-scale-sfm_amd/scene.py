"""Deterministic synthetic BA scenes.

Cameras sit on a ring of radius 10 around the origin and look at it; points are
uniform in [-1,1]^3; every point is seen by ``d`` cameras ``start + s*stride`` (mod m);
measurements are the exact projection plus U(-0.25, 0.25) px; the initial guess is the
truth perturbed (centres / rotation +-0.005, focal +-1 %, points +-0.01).

The scene is expressed in the parameterisation of Bundler's ``run_sfm``
(reference SiftGPU/ba_libs/lib/sfm-driver/sfm.c:657-709): per camera
``[C or t (3), omega (3), f*f_scale]`` around a fixed initial rotation ``R0``,
per point ``[X Y Z]``; ``proj_sign=-1`` is Bundler's -z camera (sfm.c:333),
``+1`` the VisualSFM / PBA +z camera (Optimizer/DataInterface.h:50-52).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

F_SCALE = 0.001  # sfm.c:640
K_SCALE = 5.0    # sfm.c:641

#: (cameras, points, observations-per-point) of BASELINE.json's configs C1..C3
CONFIGS = {
    "C1": (20, 2_000, 5),
    "C2": (200, 100_000, 10),
    "C3": (1_000, 1_000_000, 10),
    "micro": (5, 50, 4),
    # BAL-Ladybug-like (BASELINE config C4): ~1.7k cameras along a closed path, heavy-tailed track lengths 3 .. 120 (a few
    # longer than a wavefront), camera sets that hardly ever repeat, a wide band in S; ~3.7 M observations
    "C4": (1_700, 400_000, 120),
}
SEED_BASE = 20260


@dataclass
class Scene:
    n: int
    m: int
    cnp: int
    est_focal: int
    undistort: int
    explicit_centers: int
    proj_sign: float
    pt_ptr: np.ndarray      # int32 [n+1]
    obs_cam: np.ndarray     # int32 [nobs], cameras ascending inside a point
    x: np.ndarray           # float64 [nobs, 2]
    R0: np.ndarray          # float64 [m, 9]
    f_fixed: np.ndarray     # float64 [m]
    p0: np.ndarray          # float64 [m*cnp + 3n] initial parameters
    p_true: np.ndarray      # float64, ground truth in the same packing (omega relative to R0)
    name: str = ""
    meta: dict = field(default_factory=dict)
    nk: int = 2             # distortion parameters when undistort: 2 = k1,k2 (Bundler), 1 = k1 only
    constraints: dict | None = None   # soft constraints: cam_cw / cam_cv [m, cnp], pt_cw [n] / pt_cv [n, 3]

    @property
    def nobs(self) -> int:
        return int(self.pt_ptr[-1])

    def obs_pt(self) -> np.ndarray:
        return np.repeat(np.arange(self.n, dtype=np.int32), np.diff(self.pt_ptr))


def _rodrigues(w: np.ndarray) -> np.ndarray:
    """exp([w]x) for a batch of rotation vectors [..., 3] -> [..., 3, 3]."""
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    safe = np.where(th == 0, 1.0, th)
    k = w / safe
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s, c = np.sin(th)[..., None], np.cos(th)[..., None]
    return np.eye(3) + s * K + (1 - c) * (K @ K)


def log_so3(R: np.ndarray) -> np.ndarray:
    """rotation vector of small rotations [..., 3, 3] -> [..., 3]."""
    cos = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1) / 2, -1, 1)
    th = np.arccos(cos)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = np.sin(th)
    scale = np.where(s < 1e-12, 0.5, th / (2 * np.where(s < 1e-12, 1.0, s)))
    return v * scale[..., None]


def project(R: np.ndarray, c: np.ndarray, f: np.ndarray, X: np.ndarray, proj_sign: float,
            explicit_centers: int = 1, k1=0.0, k2=0.0) -> np.ndarray:
    """sfm_project_rd (sfm.c:303-374) for per-observation arrays R [k,3,3], c [k,3], f [k], X [k,3]."""
    q = np.einsum("kij,kj->ki", R, X - c) if explicit_centers else np.einsum("kij,kj->ki", R, X) + c
    p = proj_sign * q[:, :2] * f[:, None] / q[:, 2:3]
    rsq = (p ** 2).sum(1) / f ** 2
    return p * (1 + k1 * rsq + k2 * rsq ** 2)[:, None]


def make_scene(name: str = "C1", *, m: int | None = None, n: int | None = None, d: int | None = None,
               seed: int | None = None, proj_sign: float = -1.0, explicit_centers: int = 1,
               est_focal: int = 1, strides=(1, 2, 3), noise_px: float = 0.25,
               cam_noise: float = 0.005, f_noise: float = 0.01, pt_noise: float = 0.01,
               undistort: int = 0, nk: int = 2, k_true=(-0.25, 0.12), md_k: float = -1.25e-7,
               track_len: np.ndarray | None = None) -> Scene:
    """undistort 1: Bundler's projection distortion with nk coefficients k_true; undistort 2: PBA's measurement
    distortion x (1 + md_k |x|^2) = projection (SparseBundleCPU.cpp:1188-1193), one coefficient.
    track_len [n]: per-point number of views (<= d): the point is seen by the first track_len[i] cameras of its stride run"""
    if name == "C4" and track_len is None and m is None and n is None and d is None:
        return ladybug_like_scene(seed=seed, proj_sign=proj_sign, explicit_centers=explicit_centers, est_focal=est_focal)
    if name in CONFIGS:
        cm, cn, cd = CONFIGS[name]
        idx = list(CONFIGS).index(name)
    else:
        cm, cn, cd, idx = 20, 2000, 5, 99
    m = m or cm
    n = n or cn
    d = d or cd
    rng = np.random.default_rng(SEED_BASE + idx if seed is None else seed)
    f0 = 800.0

    # --- ground truth cameras: ring of radius 10, small height jitter, looking at the origin
    ang = 2 * np.pi * np.arange(m) / m
    C = np.stack([10 * np.cos(ang), 10 * np.sin(ang), rng.uniform(0, 0.3, m)], 1)
    fwd = -C / np.linalg.norm(C, axis=1, keepdims=True)           # viewing direction
    zc = fwd * proj_sign                                           # camera z axis (+z or -z looks)
    up = np.array([0.0, 0.0, 1.0])
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(zc, xc)
    R_true = np.stack([xc, yc, zc], 1)                             # rows = camera axes
    f_true = np.full(m, f0)

    # --- points and visibility
    X_true = rng.uniform(-1, 1, (n, 3))
    start = rng.integers(0, m, n)
    stride = rng.choice(np.asarray(strides), n)
    if track_len is not None:   # ragged tracks: flat arrays only (n x d would not fit for long-tailed lengths)
        assert m >= d * max(strides) and track_len.max() <= d and track_len.min() >= 1
        L = np.asarray(track_len, np.int64)
        pt_ptr64 = np.concatenate([[0], np.cumsum(L)])
        obs_pt = np.repeat(np.arange(n), L)
        within = np.arange(pt_ptr64[-1]) - pt_ptr64[:-1][obs_pt]
        cam = (start[obs_pt] + stride[obs_pt] * within) % m
        order = np.lexsort((cam, obs_pt))               # cameras ascending inside a point
        obs_cam = cam[order].astype(np.int32)
        pt_ptr = pt_ptr64.astype(np.int32)
    else:
        if m >= d * max(strides):
            cams = (start[:, None] + stride[:, None] * np.arange(d)[None, :]) % m
        else:  # tiny scenes: the stride pattern would wrap onto itself -> d distinct random cameras
            assert d <= m
            cams = np.argsort(rng.random((n, m)), axis=1)[:, :d]
        cams.sort(axis=1)
        obs_cam = cams.reshape(-1).astype(np.int32)
        pt_ptr = (np.arange(n + 1) * d).astype(np.int32)
        obs_pt = np.repeat(np.arange(n), d)

    # --- measurements
    Xo = X_true[obs_pt]
    t_true = -np.einsum("mij,mj->mi", R_true, C)
    c_true = C if explicit_centers else t_true
    if undistort == 2:
        nk = 1
    k1t, k2t = (k_true[0], k_true[1] if nk == 2 else 0.0) if undistort == 1 else (0.0, 0.0)
    x = project(R_true[obs_cam], c_true[obs_cam], f_true[obs_cam], Xo, proj_sign, explicit_centers, k1t, k2t)
    if undistort == 2:                             # invert x (1 + k |x|^2) = u by fixed-point iteration
        u = x
        for _ in range(50):
            x = u / (1 + md_k * (x ** 2).sum(1))[:, None]
        k1t = md_k
    x = x + rng.uniform(-noise_px, noise_px, x.shape)

    # --- perturbed initial guess
    dw = rng.uniform(-cam_noise, cam_noise, (m, 3))
    R0 = _rodrigues(dw) @ R_true
    c0 = c_true + rng.uniform(-cam_noise, cam_noise, (m, 3))
    f_init = f_true * (1 + rng.uniform(-f_noise, f_noise, m))
    X0 = X_true + rng.uniform(-pt_noise, pt_noise, (n, 3))

    cnp = 6 + (1 if est_focal else 0) + (nk if undistort else 0)
    a0 = np.zeros((m, cnp))
    a0[:, :3] = c0
    a_true = np.zeros((m, cnp))
    a_true[:, :3] = c_true
    a_true[:, 3:6] = log_so3(R_true @ np.transpose(R0, (0, 2, 1)))   # exp(w) R0 = R_true
    if est_focal:
        a0[:, 6] = f_init * F_SCALE
        a_true[:, 6] = f_true * F_SCALE
    if undistort:                                  # the initial guess has no distortion
        a_true[:, 6 + (1 if est_focal else 0)] = k1t * K_SCALE
        if nk == 2:
            a_true[:, 7 + (1 if est_focal else 0)] = k2t * K_SCALE
    p0 = np.concatenate([a0.reshape(-1), X0.reshape(-1)])
    p_true = np.concatenate([a_true.reshape(-1), X_true.reshape(-1)])
    return Scene(n=n, m=m, cnp=cnp, est_focal=int(est_focal), undistort=int(undistort), nk=int(nk),
                 explicit_centers=int(explicit_centers), proj_sign=float(proj_sign),
                 pt_ptr=pt_ptr, obs_cam=obs_cam, x=np.ascontiguousarray(x),
                 R0=np.ascontiguousarray(R0.reshape(m, 9)), f_fixed=np.ascontiguousarray(f_init),
                 p0=p0, p_true=p_true, name=name, meta=dict(d=d, seed=int(SEED_BASE + idx if seed is None else seed)))


def ladybug_like_scene(m: int | None = None, n: int | None = None, dmax: int | None = None, seed: int | None = None, **kw) -> Scene:
    """BASELINE config C4's shape: many cameras along a closed path, every point seen by a run of neighbouring cameras whose
    length is heavy-tailed (3 .. dmax, 1 % of the tracks at dmax: longer than a wavefront), so the camera sets hardly ever
    repeat and the reduced camera system is a wide band"""
    cm, cn, cd = CONFIGS["C4"]
    m, n, dmax = m or cm, n or cn, dmax or cd
    seed = SEED_BASE + list(CONFIGS).index("C4") if seed is None else seed
    rng = np.random.default_rng(seed + 1)
    L = np.minimum(2 + rng.geometric(0.14, n), dmax)
    L[rng.random(n) < 0.01] = dmax
    sc = make_scene("x", m=m, n=n, d=dmax, seed=seed, strides=(1,), track_len=L, **kw)
    sc.name = "ladybug-like"
    return sc
