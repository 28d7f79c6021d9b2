"""The cluster-merge step (include/mi_merge.h; SiftGPU/Merge3D3D.cpp:564-908): correspondences by point id, the second model
moved into the first one's frame by the GPU alignment, tracks fused, side files, and the Merge3D3D process with its
optimizer_all children.  The reference file cannot be built here (Eigen, OpenCV): the expected model is a numpy
restatement of its steps, the expected transform comes from oracle/sim3_oracle.c."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg
from helpers import model_lib, read_model, rodrigues, vsfm_scene
from test_sim3 import Sim3Oracle


class MergeConfig(C.Structure):
    _fields_ = [("rounds", C.c_int), ("device", C.c_int), ("keep_colours", C.c_int), ("reserved", C.c_int * 5)]


class MergeResult(C.Structure):
    _fields_ = [("ncorr", C.c_int64), ("scale", C.c_double), ("R", C.c_double * 9), ("t", C.c_double * 3), ("best_round", C.c_int),
                ("device_ms", C.c_double), ("ncam_first", C.c_int64), ("npt_first", C.c_int64)]


class MergeScore(C.Structure):
    _fields_ = [("ncorr", C.c_int64), ("nproj", C.c_int64), ("inliers", C.c_int64), ("mean_reproj", C.c_double), ("median_reproj", C.c_double),
                ("median_dist", C.c_double), ("mean_dist", C.c_double)]


def merge_lib():
    L = model_lib()
    dp = C.POINTER(C.c_double)
    L.mi_model_merge_with_transform.restype = C.c_void_p
    L.mi_model_merge_with_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_double, dp, dp, C.c_int, C.c_int, C.POINTER(MergeResult)]
    L.mi_model_score_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_double, dp, dp, C.POINTER(MergeScore), dp, C.c_char_p]
    L.mi_model_merge_3d3d.restype = C.c_void_p
    L.mi_model_merge_3d3d.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(MergeConfig), C.POINTER(MergeResult)]
    L.mi_model_save_merge_side_files.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.mi_model_last_error.restype = C.c_char_p
    return L


def write_cluster(path, cl):
    with open(path, "w") as fo:
        fo.write("NVM_V3_R9T\n%d\n" % len(cl["f"]))
        for j in range(len(cl["f"])):
            fo.write("%s %.12g " % (cl["names"][j], cl["f"][j]) + " ".join("%.12g" % v for v in cl["R"][j].reshape(-1)))
            fo.write(" %.12g %.12g %.12g 0 0\n" % tuple(cl["t"][j]))
        fo.write("%d\n" % len(cl["X"]))
        for i, tr in enumerate(cl["tracks"]):
            fo.write("%.12g %.12g %.12g %d %d %d %d " % (*cl["X"][i], *cl["rgb"][i], len(tr)))
            fo.write("".join("%d %d %.12g %.12g " % v for v in tr) + "\n")


def clusters_of(sc, ranges, frames, rng, est_noise=0.002):
    """sub-reconstructions of one scene: cluster c holds the cameras ranges[c] and the points at least two of them see, in its
    own similarity frame X_world = s Rg X_c + T"""
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = sc.p0[:m * cnp].reshape(m, cnp); X = sc.p0[m * cnp:].reshape(n, 3)
    R = np.einsum("mij,mjk->mik", rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3)); t = a[:, :3]
    out = []
    for (lo, hi), (s, Rg, T) in zip(ranges, frames):
        cl = dict(names=["%d" % (500 + j) for j in range(lo, hi)], f=a[lo:hi, 6] / 0.001, R=np.einsum("mij,jk->mik", R[lo:hi], Rg),
                  t=(t[lo:hi] + np.einsum("mij,j->mi", R[lo:hi], T)) / s, X=[], rgb=[], tracks=[], gid=[])
        for i in range(n):
            obs = [o for o in range(sc.pt_ptr[i], sc.pt_ptr[i + 1]) if lo <= sc.obs_cam[o] < hi]
            if len(obs) < 2:
                continue
            cl["X"].append((X[i] - T) @ Rg / s + rng.normal(0, est_noise, 3))
            cl["rgb"].append((i % 256, (2 * i) % 256, (3 * i) % 256))
            cl["tracks"].append([(int(sc.obs_cam[o]) - lo, 1000 + i, float(sc.x[o, 0]), float(sc.x[o, 1])) for o in obs])
            cl["gid"].append(i)
        cl["X"] = np.array(cl["X"])
        out.append(cl)
    return out


def np_views(cl):
    """ReadNVM, Merge3D3D.cpp:508-523: the id is the last view's field, a view on the previous view's camera is dropped"""
    ids, tracks = [], []
    for tr in cl["tracks"]:
        ids.append(tr[-1][1] if tr else -1)
        kept, prev = [], -1
        for v in tr:
            if v[0] == prev:
                continue
            prev = v[0]; kept.append(v)
        tracks.append(kept)
    return ids, tracks


def np_matches(c1, c2):
    id1, id2 = np_views(c1)[0], np_views(c2)[0]
    match = []
    for i, a in enumerate(id1):                                         # :646-683, the quadratic scan as written there
        for j, b in enumerate(id2):
            if a == b:
                match.append((i, j)); break
    return match


def np_merge(c1, c2, s, R, T, recolour=True, erase=False):
    id1, tr1 = np_views(c1); id2, tr2 = np_views(c2)
    match = np_matches(c1, c2)
    nc1, n1 = len(c1["f"]), len(c1["X"])
    X2 = (s * c2["X"]) @ R.T + T
    R2 = np.einsum("mij,kj->mik", c2["R"], R)
    cen = -np.einsum("mji,mj->mi", c2["R"], c2["t"])
    cen = (s * cen) @ R.T + T
    t2 = -np.einsum("mij,mj->mi", R2, cen)
    rgb1 = np.tile([255, 0, 0], (n1, 1)) if recolour else np.array(c1["rgb"])
    rgb2 = np.tile([0, 255, 0], (len(c2["X"]), 1)) if recolour else np.array(c2["rgb"])
    tracks = [list(t) for t in tr1]
    for i, j in match:
        if recolour:
            rgb1[i] = (0, 0, 255); rgb2[j] = (255, 255, 255)
    tr2 = [[(v[0] + nc1,) + v[1:] for v in t] for t in tr2]
    rgb2 = [tuple(c) for c in rgb2]; X2 = list(X2); id2 = list(id2)
    for i, j in sorted(match, key=lambda p: p[1])[::-1]:                 # :786-795 / MergeClusters.cpp:983-992, lists and all
        tracks[i] += tr2[j]
        if erase:
            for lst in (tr2, rgb2, X2, id2):
                del lst[j]
    tracks += tr2
    ids = id1 + id2
    rgb2 = np.array(rgb2).reshape(-1, 3); X2 = np.array(X2).reshape(-1, 3)
    cam = [v[0] for t in tracks for v in t]; pt = [i for i, t in enumerate(tracks) for _ in t]
    fid = [ids[i] for i, t in enumerate(tracks) for _ in t]; xy = [(v[2], v[3]) for t in tracks for v in t]
    return dict(match=match, f=np.concatenate([c1["f"], c2["f"]]), R=np.concatenate([c1["R"], R2]).reshape(-1, 9), t=np.concatenate([c1["t"], t2]),
                xyz=np.concatenate([c1["X"], X2]), rgb=np.concatenate([rgb1, rgb2]), cam=np.array(cam), pt=np.array(pt), fid=np.array(fid),
                xy=np.array(xy), names=c1["names"] + c2["names"])


FRAMES = [(1.0, np.eye(3), np.zeros(3)), (1.8, rodrigues(np.array([[0.4, -0.7, 0.5]]))[0], np.array([3.0, -1.0, 2.0])),
          (0.6, rodrigues(np.array([[-0.2, 0.3, 1.1]]))[0], np.array([-2.0, 0.5, 1.0]))]


def np_score(c1, c2, s, R, T):
    """MergeClusters.cpp:701-861"""
    match = np_matches(c1, c2)
    _, tr2 = np_views(c2)
    R2 = np.einsum("mij,kj->mik", c2["R"], R)
    cen = (s * -np.einsum("mji,mj->mi", c2["R"], c2["t"])) @ R.T + T
    dist, err = [], []
    for i, j in match:
        p1 = c1["X"][i]; p2 = R @ (s * c2["X"][j]) + T
        dist.append(np.linalg.norm(p2 - p1))
        for cam, _, x, y in tr2[j]:
            q = R2[cam] @ (p1 - cen[cam]); mx, my = q[0] / q[2], q[1] / q[2]
            r2 = 1 + c2.get("k1", np.zeros(len(c2["f"])))[cam] * (mx * mx + my * my)
            err.append(np.hypot(x - r2 * mx * c2["f"][cam], y - r2 * my * c2["f"][cam]))
    dist, err = np.array(dist), np.array(err)
    return dict(ncorr=len(match), nproj=len(err), mean_reproj=err.mean(), median_reproj=np.sort(err)[len(err) // 2], median_dist=np.sort(dist)[len(dist) // 2],
                mean_dist=dist.mean(), inliers=int((np.sort(dist) < np.sort(dist).sum() / len(dist)).sum()), dist=dist)


def test_candidate_scores_and_the_erasing_merge(tmp_path):
    """host only (MergeClusters.cpp): the score of a candidate transform and the fusion that erases matched points, against numpy"""
    L = merge_lib()
    sc = vsfm_scene(m=40, n=1200, d=6, seed=5, strides=(1, 2, 3), cam_noise=0.002, pt_noise=0.004, f_noise=0.002)
    rng = np.random.default_rng(1)
    c1, c2 = clusters_of(sc, [(0, 20), (20, 40)], FRAMES[:2], rng)
    c1["tracks"][2].insert(1, c1["tracks"][2][0])                       # a repeated camera
    p1, p2 = str(tmp_path / "a.nvm"), str(tmp_path / "b.nvm")
    write_cluster(p1, c1); write_cluster(p2, c2)
    h1, h2 = L.mi_model_load_nvm(p1.encode()), L.mi_model_load_nvm(p2.encode())
    s, Rg, T = FRAMES[1]
    dp = C.POINTER(C.c_double)
    for k, (sk, Rk, Tk) in enumerate([(s, Rg, T), (s * 1.01, Rg @ rodrigues(np.array([[0.01, 0, 0.02]]))[0], T + 0.05)]):
        want = np_score(c1, c2, sk, Rk, Tk)
        got = MergeScore(); dist = np.zeros(want["ncorr"]); dbg = str(tmp_path / ("debug_%d.txt" % k))
        Rc = np.ascontiguousarray(Rk.reshape(-1)); Tc = np.ascontiguousarray(Tk, np.float64)
        assert L.mi_model_score_transform(h1, h2, sk, Rc.ctypes.data_as(dp), Tc.ctypes.data_as(dp), C.byref(got), dist.ctypes.data_as(dp), dbg.encode()) == 0
        assert (got.ncorr, got.nproj, got.inliers) == (want["ncorr"], want["nproj"], want["inliers"])
        for key in ("mean_reproj", "median_reproj", "median_dist", "mean_dist"):
            assert abs(getattr(got, key) - want[key]) < 1e-9 * max(1.0, want[key]), key
        assert np.abs(dist - want["dist"]).max() < 1e-9                  # the models went through text at 12 digits
        lines = open(dbg).read().split("\n")
        assert lines[-2].startswith("ReprojErr Mean %f Median %f" % (want["mean_reproj"], want["median_reproj"]))
        assert len(lines) == 2 * want["ncorr"] + want["nproj"] + 2       # a blank line + a point line per pair, a line per view, the summary
        if k == 0:
            assert got.mean_reproj < 1.0 and got.mean_dist < 0.01        # the true transform: the estimates' noise
        else:
            assert got.mean_reproj > 1.0 and got.mean_dist > 0.02
    # the fusion with a given transform, matched points of the second model erased
    res = MergeResult()
    Rc = np.ascontiguousarray(Rg.reshape(-1)); Tc = np.ascontiguousarray(T, np.float64)
    h = L.mi_model_merge_with_transform(h1, h2, s, Rc.ctypes.data_as(dp), Tc.ctypes.data_as(dp), 1, 0, C.byref(res))
    got = read_model(L, h)
    ref = np_merge(c1, c2, s, Rg, T, erase=True)
    assert res.ncorr == len(ref["match"]) > 100 and len(got["xyz"]) == len(c1["X"]) + len(c2["X"]) - res.ncorr
    assert np.array_equal(got["cam"], ref["cam"]) and np.array_equal(got["pt"], ref["pt"]) and np.array_equal(got["fid"], ref["fid"])
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.abs(got["xy"] - ref["xy"]).max() < 1e-9
    assert np.abs(got["R"] - ref["R"]).max() < 1e-9 and np.abs(got["t"] - ref["t"]).max() < 1e-8 and np.abs(got["xyz"] - ref["xyz"]).max() < 1e-8
    assert L.mi_model_rmse(h) < 1.0                                      # one model in one frame: only the estimates' noise is left
    L.mi_model_free(h)
    # not erased = Merge3D3D's fusion with the same transform
    h = L.mi_model_merge_with_transform(h1, h2, s, Rc.ctypes.data_as(dp), Tc.ctypes.data_as(dp), 0, 1, None)
    got = read_model(L, h); ref = np_merge(c1, c2, s, Rg, T, recolour=False)
    assert np.array_equal(got["cam"], ref["cam"]) and np.array_equal(got["rgb"], ref["rgb"]) and np.abs(got["xyz"] - ref["xyz"]).max() < 1e-8
    L.mi_model_free(h)
    # two points of the first model with one partner: the later one takes -- and removes -- the next surviving track (the reference's erase by position)
    c1d = dict(c1); c1d["tracks"] = [list(t) for t in c1["tracks"]]
    i_a, j_a = ref["match"][10]
    i_b = [i for i in range(len(c1["X"])) if i not in dict(ref["match"])][0]
    c1d["tracks"][i_b] = [(v[0], c1["tracks"][i_a][-1][1], v[2], v[3]) for v in c1["tracks"][i_b]]
    p1d = str(tmp_path / "ad.nvm"); write_cluster(p1d, c1d)
    h1d = L.mi_model_load_nvm(p1d.encode())
    h = L.mi_model_merge_with_transform(h1d, h2, s, Rc.ctypes.data_as(dp), Tc.ctypes.data_as(dp), 1, 0, C.byref(res))
    got = read_model(L, h)
    assert res.ncorr == len(ref["match"]) + 1 and len(got["xyz"]) == len(c1["X"]) + len(c2["X"]) - res.ncorr
    nv = lambda i: int((got["pt"] == i).sum())
    own = lambda i: len(np_views(c1d)[1][i])
    tr2 = np_views(c2)[1]
    matched2 = sorted(j for _, j in ref["match"])
    nxt = next(j for j in range(j_a + 1, len(tr2)) if j not in matched2 or j > j_a)    # the element that slides into position j_a
    assert sorted([nv(i_a) - own(i_a), nv(i_b) - own(i_b)]) == sorted([len(tr2[j_a]), len(tr2[nxt])])
    for x in (h, h1, h2, h1d):
        L.mi_model_free(x)


def test_side_files_of_a_model(tmp_path):
    """host only: the files the reference leaves beside the merged NVM, formats of Merge3D3D.cpp:811-900"""
    L = merge_lib()
    sc = vsfm_scene(m=6, n=40, d=3, seed=2)
    rng = np.random.default_rng(0)
    cl = clusters_of(sc, [(0, 6)], FRAMES[:1], rng)[0]
    nvm = str(tmp_path / "m.nvm"); write_cluster(nvm, cl)
    h = L.mi_model_load_nvm(nvm.encode())
    assert L.mi_model_save_merge_side_files(h, str(tmp_path).encode(), 5) == 0
    L.mi_model_free(h)
    n = len(cl["X"])
    ply = open(tmp_path / "merged.ply").read().split("\n")
    assert ply[0] == "ply" and ply[1] == "format ascii 1.0" and ply[2].split() == ["element", "vertex", str(n + 5)]
    assert ply[3] == "property float x" and ply[9] == "end_header" and len(ply) == 10 + n + 5 + 1
    assert ply[10 + n:10 + n + 5] == ply[10 + n - 5:10 + n]              # the tail is listed twice
    foc = open(tmp_path / "focal_for_optimisation.txt").read().split("\n")
    assert foc[0] == "6" and foc[1] == "500 %.9f 2" % cl["f"][0]
    assert open(tmp_path / "Invmap.txt").read().split("\n")[5] == "5 505"
    rt = np.loadtxt(tmp_path / "RTglobalmapped_for_optimisation.txt", max_rows=1)
    assert np.abs(rt - cl["R"][0].reshape(-1)).max() < 1e-8
    cen = np.array([float(v) for v in open(tmp_path / "RTglobalmapped_for_optimisation.txt").read().split("\n")[1].split()])
    assert np.abs(cen + cl["R"][0].T @ cl["t"][0]).max() < 1e-8
    ours = open(tmp_path / "ours_new.txt").read().split("\n")[0].split()
    tr = cl["tracks"][0]
    assert int(ours[6]) == len(tr) and [int(ours[7]), int(ours[8])] == [tr[0][0], tr[0][1]] and abs(float(ours[9]) - tr[0][2]) < 1e-6
    assert L.mi_model_save_merge_side_files(None, b"/tmp", 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("keep_colours", [0, 1])
def test_merge_matches_the_restated_steps(tmp_path, keep_colours):
    L = merge_lib()
    sc = vsfm_scene(m=60, n=3000, d=6, seed=123, strides=(1, 2, 3), cam_noise=0.002, pt_noise=0.004, f_noise=0.002)
    rng = np.random.default_rng(9)
    c1, c2 = clusters_of(sc, [(0, 30), (30, 60)], FRAMES[:2], rng)
    # the reference's quirks: a repeated camera in a track, an id taken from the last view, an id that occurs twice in the second model
    c1["tracks"][3].insert(1, c1["tracks"][3][0])
    common = sorted(set(c1["gid"]) & set(c2["gid"]))
    assert len(common) > 300
    j_a, j_b = c2["gid"].index(common[5]), c2["gid"].index(common[9])
    c2["tracks"][j_b] = [(v[0], 1000 + common[5], v[2], v[3]) for v in c2["tracks"][j_b]]       # j_b now carries common[5]'s id too
    k = c2["gid"].index(common[20])
    c2["tracks"][k][-1] = (c2["tracks"][k][-1][0], 77, c2["tracks"][k][-1][2], c2["tracks"][k][-1][3])   # its id becomes 77: no partner
    p1, p2 = str(tmp_path / "a.nvm"), str(tmp_path / "b.nvm")
    write_cluster(p1, c1); write_cluster(p2, c2)
    h1, h2 = L.mi_model_load_nvm(p1.encode()), L.mi_model_load_nvm(p2.encode())
    cfg = MergeConfig(rounds=300, device=0, keep_colours=keep_colours); res = MergeResult()
    h = L.mi_model_merge_3d3d(h1, h2, C.byref(cfg), C.byref(res))
    assert h, L.mi_model_last_error()
    got = read_model(L, h)
    # the transform: the GPU alignment against the CPU restatement on the same correspondences (second -> first)
    ref = np_merge(c1, c2, res.scale, np.array(res.R).reshape(3, 3), np.array(res.t), recolour=not keep_colours)
    assert res.ncorr == len(ref["match"]) == len(common) - 2            # the id-77 point and the shadowed duplicate drop out
    assert (min(j_a, j_b) in [j for _, j in ref["match"]]) and (max(j_a, j_b) not in [j for _, j in ref["match"]])
    src = np.array([c2["X"][j] for _, j in ref["match"]]); dst = np.array([c1["X"][i] for i, _ in ref["match"]])
    o = Sim3Oracle().align(src, dst, 300)
    assert res.best_round == o["best_round"] and abs(res.scale - o["scale"]) < 1e-12 * o["scale"]
    assert np.abs(np.array(res.R).reshape(3, 3) - o["R"]).max() < 1e-9 and np.abs(np.array(res.t) - o["t"]).max() < 1e-8
    assert abs(res.scale - 1.8) < 0.01 and (res.ncam_first, res.npt_first) == (30, len(c1["X"]))
    # the merged model
    assert np.array_equal(got["cam"], ref["cam"]) and np.array_equal(got["pt"], ref["pt"]) and np.array_equal(got["fid"], ref["fid"])
    assert np.array_equal(got["rgb"], ref["rgb"]) and np.abs(got["xy"] - ref["xy"]).max() < 1e-9
    assert np.abs(got["f"] - ref["f"]).max() < 1e-9 and np.abs(got["R"] - ref["R"]).max() < 1e-9
    assert np.abs(got["t"] - ref["t"]).max() < 1e-8 and np.abs(got["xyz"] - ref["xyz"]).max() < 1e-8
    assert len(got["xyz"]) == len(c1["X"]) + len(c2["X"])               # the matched points of the second model stay as well
    # a similarity moves cameras and points together: the second model's own views keep their residuals
    n1 = len(c1["X"])
    own = got["pt"] >= n1
    Rm = got["R"].reshape(-1, 3, 3)[got["cam"][own]]; Xm = got["xyz"][got["pt"][own]]
    q = np.einsum("kij,kj->ki", Rm, Xm) + got["t"][got["cam"][own]]
    res_px = got["xy"][own] - got["f"][got["cam"][own], None] * q[:, :2] / q[:, 2:]
    assert abs(np.sqrt((res_px ** 2).sum() / own.sum()) - L.mi_model_rmse(h2)) < 1e-6
    nvm = str(tmp_path / "m.nvm")
    assert L.mi_model_save_nvm(h, nvm.encode()) == 0
    for x in (h, h1, h2):
        L.mi_model_free(x)
    # too few common points is an error, not a crash
    c3 = dict(c2); c3["tracks"] = [[(v[0], 5, v[2], v[3]) for v in t] for t in c2["tracks"]]
    p3 = str(tmp_path / "c.nvm"); write_cluster(p3, c3)
    h1, h3 = L.mi_model_load_nvm(p1.encode()), L.mi_model_load_nvm(p3.encode())
    assert not L.mi_model_merge_3d3d(h1, h3, C.byref(cfg), C.byref(res)) and b"common points" in L.mi_model_last_error()
    L.mi_model_free(h1); L.mi_model_free(h3)


@pytest.mark.gpu
def test_merge3d3d_process_merges_three_clusters_and_refines(tmp_path):
    """Merge3D3D <dir>: dendogram/fullclustermergeinfo.txt names two merges (D = A + B, E = D + C); every merged cluster gets its
    NVM and side files and is bundle adjusted by the optimizer_all child; the final model holds all 90 cameras at the noise level"""
    L = merge_lib()
    sc = vsfm_scene(m=90, n=5000, d=6, seed=31, strides=(1, 2, 3), cam_noise=0.002, pt_noise=0.004, f_noise=0.002)
    rng = np.random.default_rng(3)
    cls = clusters_of(sc, [(0, 30), (30, 60), (60, 90)], FRAMES, rng)
    den = tmp_path / "dendogram"
    for name, cl in zip("ABC", cls):
        (den / ("cluster" + name)).mkdir(parents=True)
        write_cluster(str(den / ("cluster" + name) / "outputVSFM_GB.nvm"), cl)
    (den / "fullclustermergeinfo.txt").write_text("D A B\nE D C\n")
    exe = os.path.join(ROOT, "large-scale-sfm_amd", "bin", "Merge3D3D")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("#Correspondences") == 2 and out.stdout.count("LM iterations") == 2
    for name in "DE":
        for fn in ("outputVSFM_GB.nvm", "RTglobalmapped_for_optimisation.txt", "focal_for_optimisation.txt", "Invmap.txt", "ours_new.txt", "merged.ply"):
            assert (den / ("cluster" + name) / fn).stat().st_size > 0
    h = L.mi_model_load_nvm(str(den / "clusterE" / "outputVSFM_GB.nvm").encode())
    mdl = read_model(L, h)
    assert len(mdl["f"]) == 90 and L.mi_model_rmse(h) < 0.25            # three frames, two seams, one model at the noise level
    assert np.all(np.abs(mdl["f"] - 800) < 40)
    L.mi_model_free(h)
    assert [l.split()[1] for l in open(den / "clusterE" / "Invmap.txt").read().split("\n")[:90]] == ["%d" % (500 + j) for j in range(90)]
    # a missing list is an error message, not a crash
    bad = subprocess.run([exe, str(tmp_path / "nowhere")], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


def _write_rts(path, s, R, T):
    with open(path, "w") as fo:
        fo.write("%.12g\n" % s + " ".join("%.12g" % v for v in R.reshape(-1)) + "\n" + " ".join("%.12g" % v for v in T) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("good", [True, False])
def test_mergeclusters_process_takes_the_best_candidate_and_refines(tmp_path, good):
    """MergeClusters <dir> <index>: the candidate transforms other stage programs left in the new cluster's directory are scored, the
    one with the smallest mean distance fuses the models (matched points erased), and optimizer_all runs with intrinsics held for 5
    iterations when that candidate reprojects within 10 pixels, otherwise as a full 50-iteration bundle (MergeClusters.cpp:1102-1108)"""
    L = merge_lib()
    sc = vsfm_scene(m=60, n=3000, d=6, seed=77, strides=(1, 2, 3), cam_noise=0.002, pt_noise=0.004, f_noise=0.002)
    rng = np.random.default_rng(4)
    c1, c2 = clusters_of(sc, [(0, 30), (30, 60)], FRAMES[:2], rng)
    den = tmp_path / "dendogram"
    for name, cl in zip("AB", (c1, c2)):
        (den / ("cluster" + name)).mkdir(parents=True)
        write_cluster(str(den / ("cluster" + name) / "outputVSFM_GB.nvm"), cl)
    (den / "fullclustermergeinfo.txt").write_text("X P Q\nD A B\n")
    (den / "clusterD").mkdir()
    s, Rg, T = FRAMES[1]
    bend = lambda a: (s * (1 + a), Rg @ rodrigues(np.array([[a, -a, 2 * a]]))[0], T + 3 * a)
    cands = {"RTS2D2D.txt": bend(0.06), "RTS3D3D.txt": bend(0.02 if good else 0.045), "RTSRefined.txt": bend(0.0 if good else 0.07), "RTSRefined_1.txt": bend(0.055)}
    for fn, g in cands.items():                                          # RTS2DR3DT.txt is missing: skipped with a message
        _write_rts(str(den / "clusterD" / fn), *g)
    exe = os.path.join(ROOT, "large-scale-sfm_amd", "bin", "MergeClusters")
    out = subprocess.run([exe, str(tmp_path), "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    score_lines = [l.split() for l in out.stdout.splitlines() if l.startswith("######")]
    assert [int(l[1]) for l in score_lines] == [0, 1, 3, 4] and "RTS2DR3DT.txt" in out.stderr
    best = min(score_lines, key=lambda l: float(l[5]))
    assert int(best[1]) == (3 if good else 1)
    want = np_score(c1, c2, *cands["RTSRefined.txt" if good else "RTS3D3D.txt"])
    assert abs(float(best[2]) - want["mean_reproj"]) < 1e-4 and abs(float(best[5]) - want["mean_dist"]) < 1e-5 and int(best[6]) == want["inliers"]
    assert (want["mean_reproj"] > 10.0) == (not good)
    for fn in ("outputVSFM_GB.nvm", "RTglobalmapped_for_optimisation.txt", "focal_for_optimisation.txt", "Invmap.txt", "ours_new.txt", "merged.ply",
               "debug_0.txt", "debug_err.txt"):
        assert (den / "clusterD" / fn).stat().st_size > 0
    err = np.loadtxt(den / "clusterD" / "debug_err.txt")
    assert err.shape == (want["ncorr"], 4) and np.all(err[:, 2] == 0) and np.abs(err[:, 3 if good else 1] - want["dist"]).max() < 1e-5
    h = L.mi_model_load_nvm(str(den / "clusterD" / "outputVSFM_GB.nvm").encode())
    mdl = read_model(L, h); rmse = L.mi_model_rmse(h); L.mi_model_free(h)
    assert len(mdl["f"]) == 60 and len(mdl["xyz"]) == len(c1["X"]) + len(c2["X"]) - want["ncorr"]
    f_in = np.concatenate([c1["f"], c2["f"]])
    lm = [l for l in out.stdout.splitlines() if "LM iterations" in l][0]
    if good:                                                             # "-1 5": intrinsics held, at most 5 iterations
        assert np.abs(mdl["f"] - f_in).max() < 1e-8 and int(lm.split()[1]) <= 5 and rmse < 0.5
    else:                                                                # "0 50": the full bundle closes the seam
        assert np.abs(mdl["f"] - f_in).max() > 1e-3 and rmse < 0.25
    # an index past the list does nothing; a missing directory is an error
    assert subprocess.run([exe, str(tmp_path), "7"], capture_output=True, text=True, timeout=60).returncode == 0
    assert subprocess.run([exe, str(tmp_path / "nowhere"), "0"], capture_output=True, text=True, timeout=60).returncode == 1


@pytest.mark.gpu
def test_config_c5_eight_clusters_through_the_stage_programs(tmp_path):
    """BASELINE config C5 as the pipeline does it: 8 sub-reconstructions of one scene, each in its own similarity frame, fused
    pairwise up the dendrogram by Merge3D3D (7 alignments on the GPU, 7 bundles by its optimizer_all children), then the merged
    model refined once more with the points sharded over 3 ranks (MI_BA_GPUS; on this one-GPU box through the host transport)"""
    L = merge_lib()
    sc = vsfm_scene(m=160, n=12000, d=6, seed=55, strides=(1, 2, 3), cam_noise=0.002, pt_noise=0.004, f_noise=0.002)
    rng = np.random.default_rng(8)
    frames = [FRAMES[0]] + [(float(np.exp(rng.normal(0, 0.4))), rodrigues(rng.normal(0, 0.6, (1, 3)))[0], rng.normal(0, 2.0, 3)) for _ in range(7)]
    cls = clusters_of(sc, [(20 * c, 20 * c + 20) for c in range(8)], frames, rng)
    den = tmp_path / "dendogram"
    for name, cl in zip("ABCDEFGH", cls):
        (den / ("cluster" + name)).mkdir(parents=True)
        write_cluster(str(den / ("cluster" + name) / "outputVSFM_GB.nvm"), cl)
    (den / "fullclustermergeinfo.txt").write_text("I A B\nJ C D\nK E F\nL G H\nM I J\nN K L\nO M N\n")
    exe = os.path.join(ROOT, "large-scale-sfm_amd", "bin", "Merge3D3D")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("#Correspondences") == 7 and out.stdout.count("LM iterations") == 7
    nvm = str(den / "clusterO" / "outputVSFM_GB.nvm")
    h = L.mi_model_load_nvm(nvm.encode()); mdl = read_model(L, h); rmse1 = L.mi_model_rmse(h); L.mi_model_free(h)
    assert len(mdl["f"]) == 160 and rmse1 < 0.25 and np.all(np.abs(mdl["f"] - 800) < 40)
    names = [l.split()[1] for l in open(den / "clusterO" / "Invmap.txt").read().split("\n")[:160]]
    assert sorted(names) == sorted("%d" % (500 + j) for j in range(160))
    # the global refine on "the node": one process spawned as ever, three ranks behind it
    opt = os.path.join(ROOT, "large-scale-sfm_amd", "bin", "optimizer_all")
    out3 = subprocess.run([opt, str(den / "clusterO") + "/", "0", "100", "16", "32", "0", "0"], capture_output=True, text=True, timeout=600,
                          env={**os.environ, "MI_BA_GPUS": "3", "MI_BA_TRANSPORT": "host"})
    assert out3.returncode == 0 and "3 ranks" in out3.stdout, out3.stderr
    h = L.mi_model_load_nvm(nvm.encode()); rmse2 = L.mi_model_rmse(h); L.mi_model_free(h)
    assert rmse2 <= rmse1 + 1e-6 and rmse2 < 0.21
