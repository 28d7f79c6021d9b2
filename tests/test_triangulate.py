"""The triangulation step (include/mi_triangulate.h): CPU restatement vs the reference's own triangulate_n and the real
MINPACK (golden vectors), and the HIP kernel vs both."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, pkg
from helpers import tri_scene


class TriOracle:
    def __init__(self):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
        self.L = C.CDLL(os.path.join(ROOT, "oracle", "libba_oracle.so"))
        self.L.tri_tracks.restype = C.c_double

    def run(self, ts, min_views=2, min_angle_deg=0.5, max_error=0.01, float_normalise=1, polish=0):
        n = ts["n"]
        xyz = np.zeros((n, 3)); err = np.zeros(n); status = np.zeros(n, np.int32)
        counts = (C.c_longlong * 4)(); npj = C.c_longlong(0)
        P = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
        keep = [np.ascontiguousarray(ts[k]) for k in ("f", "R", "t", "c", "pt_ptr", "obs_cam", "xy")]
        s = self.L.tri_tracks(ts["m"], *[P(k) for k in keep[:4]], n, *[P(k) for k in keep[4:]], min_views, C.c_double(min_angle_deg),
                              C.c_double(max_error), float_normalise, polish, P(xyz), P(err), P(status), counts, C.byref(npj))
        return dict(xyz=xyz, err=err, status=status, counts=list(counts), sum_sq_px=s, nproj=npj.value)


@pytest.fixture(scope="module")
def tri_oracle():
    return TriOracle()


@pytest.fixture(scope="module")
def scene_and_golden():
    ts = tri_scene()
    z = np.load(os.path.join(GOLDEN, "ref_tri.npz"))
    assert abs(np.abs(ts["xy"]).sum() - float(z["xysum"])) < 1e-9 * float(z["xysum"])
    return ts, z


def usable_views(ts):
    return np.array([(ts["obs_cam"][ts["pt_ptr"][i]:ts["pt_ptr"][i + 1]] >= 0).sum() for i in range(ts["n"])])


def test_restatement_matches_the_references_triangulate_n(tri_oracle, scene_and_golden):
    ts, z = scene_and_golden
    r = tri_oracle.run(ts, min_angle_deg=0.0, max_error=1e300)      # triangulate everything the golden run did
    nv = usable_views(ts)
    tri = z["err_ref"] >= 0
    assert np.array_equal(tri, r["status"] == 3) and np.array_equal(nv >= 2, tri)
    scale = np.maximum(1.0, np.abs(z["X_ref"]).max(1))
    d = np.abs(r["xyz"] - z["X_ref"]).max(1)[tri] / scale[tri]                       # normal equations vs dgelsy:
    assert np.quantile(d, 0.98) < 1e-12 and d.max() < 1e-6                           # rounding; the far points are ill-conditioned
    # Bundler's triangulate_n reports sqrt(sum / n), the pipeline's copy sqrt(sum) / n (TriangulationMultiStep.cpp:661)
    ok = tri & (r["err"] < 1.0)                                  # (the pipeline's copy also has the depth test -> 1e7)
    assert np.allclose(r["err"][ok] * np.sqrt(nv[ok]), z["err_ref"][ok], rtol=1e-6, atol=1e-12)


def test_lmdif_restatement_matches_minpack(tri_oracle, scene_and_golden):
    ts, z = scene_and_golden
    r = tri_oracle.run(ts, min_angle_deg=0.0, max_error=1e300, polish=1)
    tri = z["err_ref"] >= 0
    d = np.abs(r["xyz"] - z["X_minpack"]).max(1)[tri] / np.maximum(1.0, np.abs(z["X_minpack"]).max(1))[tri]
    # the two differ by rounding in a forward-difference Jacobian (h ~ 1.5e-8) and QR vs normal equations
    assert np.quantile(d, 0.99) < 1e-6 and d.max() < 1e-3
    moved = np.abs(z["X_minpack"] - z["X_ref"]).max(1)[tri]
    assert moved.max() > 1e-4                                    # the polish does something the reference build never sees


def test_track_filters(tri_oracle, scene_and_golden):
    """views of absent cameras are skipped, short tracks and narrow-angle tracks stay (0,0,0), outliers are dropped"""
    ts, z = scene_and_golden
    r = tri_oracle.run(ts)
    nv = usable_views(ts)
    n = ts["n"]
    assert np.array_equal(r["status"] == 0, nv < 2)
    far = np.arange(13, n, 97); far = far[nv[far] >= 2]
    assert np.all(np.isin(r["status"][far], (1, 2)))             # rays too parallel, or the point behind some camera
    out = np.arange(7, n, 53); out = out[(nv[out] >= 2) & (ts["obs_cam"][ts["pt_ptr"][out]] >= 0) & (np.isin(out, far, invert=True))]
    assert np.all(r["status"][out] == 2)
    kept = r["status"] == 3
    assert kept.sum() > 0.85 * n and np.all(r["xyz"][~kept] == 0.0)
    assert np.abs(r["xyz"][kept] - ts["X_true"][kept]).max() < 0.05
    assert r["counts"] == [int((nv >= 2).sum()), int((r["status"] >= 2).sum()), int(kept.sum()), int(kept.sum())]
    assert r["nproj"] == int(nv[kept].sum())
    first = tri_oracle.run(ts, min_views=3, min_angle_deg=1.0)     # the first level's thresholds (:952, :969)
    assert np.all(first["status"][nv == 2] == 0) and first["counts"][2] <= r["counts"][2]


def narrow_and_behind_scene():
    """two cameras 0.02 apart looking down +z: a point 50 away (rays 0.023 degrees apart), the same point seen by a
    third camera 5 to the side (well conditioned), and a point behind the cameras"""
    f = np.full(3, 800.0); R = np.tile(np.eye(3).reshape(-1), (3, 1)); c = np.array([[0.0, 0, 0], [0.02, 0, 0], [5.0, 0, 0]])
    t = -c.copy()
    X = np.array([[0.3, 0.2, 50.0], [0.3, 0.2, 50.0], [0.1, 0.1, -4.0]])
    tracks = [(0, [0, 1]), (1, [0, 1, 2]), (2, [0, 2])]
    cam = []; xy = []; ptr = [0]
    for i, cs in tracks:
        for j in cs:
            q = X[i] - c[j]
            cam.append(j); xy.append(f[j] * q[:2] / q[2])
        ptr.append(len(cam))
    return dict(f=f, R=R, t=t, c=c, pt_ptr=np.array(ptr, np.int32), obs_cam=np.array(cam, np.int32), xy=np.array(xy), m=3, n=3, X_true=X)


def test_narrow_angle_and_negative_depth(tri_oracle):
    ts = narrow_and_behind_scene()
    r = tri_oracle.run(ts)
    assert list(r["status"]) == [1, 3, 2]                        # narrow; kept; depth < 1e-4 -> error 1e7 (:655-658)
    assert np.all(r["xyz"][0] == 0) and np.abs(r["xyz"][1] - ts["X_true"][1]).max() < 1e-3 and r["err"][2] > 1000


# ------------------------------------------------------------------------------------------------ GPU
def gpu_run(ts, **kw):
    tri = pkg("capi").Triangulator(**kw)
    tri.set_cameras(ts["f"], ts["R"], ts["t"], ts["c"])
    tri.set_tracks(ts["pt_ptr"], ts["obs_cam"], ts["xy"])
    ms = tri.run()
    out = tri.get()
    out["ms"] = ms; out["bytes"] = tri.algorithmic_bytes()
    tri.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("polish", [0, 1])
@pytest.mark.parametrize("level", ["second", "first"])
def test_kernel_matches_the_restatement(tri_oracle, scene_and_golden, polish, level):
    ts, z = scene_and_golden
    kw = dict(min_views=2, min_angle_deg=0.5) if level == "second" else dict(min_views=3, min_angle_deg=1.0)
    want = tri_oracle.run(ts, polish=polish, **kw)
    got = gpu_run(ts, polish=polish, **kw)
    assert np.array_equal(got["status"], want["status"]) and got["counts"] == want["counts"] and got["nproj"] == want["nproj"]
    tol = 1e-11 if polish == 0 else 1e-7                  # the polish differentiates numerically: rounding is amplified
    assert np.abs(got["xyz"] - want["xyz"]).max() < tol * max(1.0, np.abs(want["xyz"]).max())
    tri = want["err"] >= 0
    assert np.allclose(got["err"][tri], want["err"][tri], rtol=1e-6, atol=1e-13) and np.all(got["err"][~tri] == -1.0)
    assert abs(got["sum_sq_px"] - want["sum_sq_px"]) < 1e-6 * want["sum_sq_px"]


@pytest.mark.gpu
def test_kernel_matches_the_reference_and_minpack(scene_and_golden):
    ts, z = scene_and_golden
    tri = z["err_ref"] >= 0
    got = gpu_run(ts, min_angle_deg=0.0, max_error=1e300)
    assert np.array_equal(got["status"] == 3, tri)
    d = np.abs(got["xyz"] - z["X_ref"]).max(1)[tri] / np.maximum(1.0, np.abs(z["X_ref"]).max(1))[tri]
    assert np.quantile(d, 0.98) < 1e-12 and d.max() < 1e-6       # the far points are ill-conditioned
    got = gpu_run(ts, min_angle_deg=0.0, max_error=1e300, polish=1)
    d = np.abs(got["xyz"] - z["X_minpack"]).max(1)[tri] / np.maximum(1.0, np.abs(z["X_minpack"]).max(1))[tri]
    assert np.quantile(d, 0.99) < 1e-6 and d.max() < 1e-3


@pytest.mark.gpu
def test_kernel_narrow_angle_and_negative_depth(tri_oracle):
    ts = narrow_and_behind_scene()
    got, want = gpu_run(ts), tri_oracle.run(ts)
    assert list(got["status"]) == [1, 3, 2] and np.allclose(got["xyz"], want["xyz"], atol=1e-12)
    assert np.allclose(got["err"], want["err"], rtol=1e-9)


@pytest.mark.gpu
def test_one_shot_call_and_errors(scene_and_golden):
    ts, z = scene_and_golden
    capi = pkg("capi")
    L = capi.lib()
    n = ts["n"]
    xyz = np.zeros((n, 3)); counts = (C.c_int64 * 4)()
    P = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    L.mi_triangulate_tracks.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 9
    keep = [np.ascontiguousarray(ts[k]) for k in ("f", "R", "t", "c", "pt_ptr", "obs_cam", "xy")]
    rc = L.mi_triangulate_tracks(None, ts["m"], *[P(k) for k in keep[:4]], n, *[P(k) for k in keep[4:]], P(xyz), None, None, counts, None, None)
    assert rc == 0 and counts[2] > 0.85 * n
    assert np.array_equal(xyz, gpu_run(ts)["xyz"])
    tri = capi.Triangulator()
    with pytest.raises(RuntimeError, match="set_cameras first"):
        tri.set_tracks(ts["pt_ptr"], ts["obs_cam"], ts["xy"])
    tri.set_cameras(ts["f"], ts["R"], ts["t"])                   # centres computed as -R^T t
    bad = ts["obs_cam"].copy(); bad[5] = ts["m"]
    with pytest.raises(RuntimeError, match="out of range"):
        tri.set_tracks(ts["pt_ptr"], bad, ts["xy"])
    with pytest.raises(RuntimeError, match="must be set first"):
        tri.run()
    tri.set_tracks(np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))     # no tracks at all
    tri.run()
    assert tri.get()["counts"] == [0, 0, 0, 0]


@pytest.mark.gpu
def test_full_size_tracks_recover_the_points():
    """10 M views / 1 M tracks (the C3 scene's tracks at the true cameras): exact measurements give back the points"""
    scene = pkg("scene")
    sc = scene.make_scene("C3", proj_sign=1.0, explicit_centers=0, noise_px=0.0)
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = sc.p_true[:m * cnp].reshape(m, cnp)
    R = np.einsum("mij,mjk->mik", scene._rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3))
    ts = dict(f=a[:, 6] / 0.001, R=R.reshape(m, 9), t=a[:, :3].copy(), c=None, pt_ptr=sc.pt_ptr.astype(np.int32),
              obs_cam=sc.obs_cam.astype(np.int32), xy=sc.x, m=m, n=n)
    tri = pkg("capi").Triangulator(float_normalise=0)
    tri.set_cameras(ts["f"], ts["R"], ts["t"])
    tri.set_tracks(ts["pt_ptr"], ts["obs_cam"], ts["xy"])
    ms = min(tri.run() for _ in range(3))
    out = tri.get()
    X = sc.p_true[m * cnp:].reshape(n, 3)
    assert out["counts"][2] == n and np.abs(out["xyz"] - X).max() < 1e-9
    print("triangulation: %.3f ms for %d tracks / %d views = %.1f M tracks/s, %.0f GB/s algorithmic" %
          (ms, n, sc.nobs, n / ms / 1e3, tri.algorithmic_bytes() / ms / 1e6))
