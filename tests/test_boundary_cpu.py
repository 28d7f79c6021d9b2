"""CPU-only checks of the drop-in boundary: the library loads, exports every symbol the headers
declare (no compute without a GPU), the file formats round-trip, and the shard planner's tables
are consistent."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, pkg
from helpers import model_lib, read_model, vsfm_scene, write_nvm, write_side_files


def test_library_exports_every_declared_symbol():
    L = pkg("capi").lib()
    names = set()
    for h in ("mi_ba.h", "mi_model.h", "mi_sfm.h", "mi_triangulate.h", "mi_sim3.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(mi_[a-z_0-9]+|run_sfm)\s*\(", txt))
    names |= {"NewParallelBA", "ParallelBA_GetVersion"}
    assert len(names) > 40
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    L.ParallelBA_GetVersion.restype = C.c_int
    assert L.ParallelBA_GetVersion() == 105            # pba.cpp:151-154


def test_timing_block_keeps_its_size():
    """mi_ba_timing is 32 doubles from round 3 on (it grew twice before it got a reserved tail); mi_ba_get_timing_n serves callers
    compiled against a shorter struct and refuses a null engine"""
    capi = pkg("capi")
    assert C.sizeof(capi.Timing) == 32 * 8
    hdr = open(os.path.join(ROOT, "include", "mi_ba.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} mi_ba_timing;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ndbl = sum(int(mm.group(1)) if (mm := re.search(r"\[(\d+)\]", f)) else 1 for decl in re.findall(r"double([^;]*);", body) for f in decl.split(","))
    assert ndbl == 32
    L = capi.lib()
    L.mi_ba_get_timing_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    buf = (C.c_double * 4)()
    assert L.mi_ba_get_timing_n(None, buf, 32) == -1


def test_parallelba_has_the_reference_cxx_abi():
    """mangled names of class ParallelBA = those of the reference's own libpba (symbol table data)"""
    import subprocess
    want = set(open(os.path.join(GOLDEN, "ref_pba_symbols.txt")).read().split())
    out = subprocess.check_output(["nm", "-D", pkg("capi").LIB_PATH]).decode()
    have = {l.split()[-1] for l in out.splitlines() if " T _ZN10ParallelBA" in l}
    assert want <= have, sorted(want - have)


def test_the_references_own_parallelba_runs_behind_our_header(tmp_path):
    """tests/native/pba_plain_client.cpp compiled against include/mi_pba.h and linked to the REFERENCE's PBA (CPU
    build, oracle/_ref/libpba_ref.so): the header is binary compatible (vtable order, CameraT / Point3D / Point2D
    layouts), and the committed golden vectors are what that library produces"""
    from helpers import build_pba_client, pba_case, run_pba_client, write_pba_problem
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libpba_ref.so")):
        pytest.skip("oracle/_ref/libpba_ref.so not built (needs /root/reference)")
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_pba.npz"))
    exe = build_pba_client(tmp_path, "ref")
    for name in ("meas_dist", "motion_only"):
        sc, kw = pba_case(name)
        prob = str(tmp_path / (name + ".txt"))
        write_pba_problem(prob, sc, **kw)
        r = run_pba_client(exe, prob, tmp_path, -3, sc.m, sc.n)
        assert r["iters"] == int(z[name + "_iters"]) and abs(r["mse"] - float(z[name + "_mse"])) < 1e-6 * r["mse"]
        assert np.abs(r["f"] - z[name + "_f"]).max() < 1e-3 and np.abs(r["pts"] - z[name + "_pts"]).max() < 1e-6


def test_camera_rotation_helpers(tmp_path):
    """CameraT_'s quaternion / Rodrigues / inverted-frame helpers (DataInterface.h:93-322) against scipy"""
    import subprocess
    from scipy.spatial.transform import Rotation
    exe = str(tmp_path / "camera_methods")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "camera_methods.cpp"), "-o", exe])
    rng = np.random.default_rng(3)
    rots = [Rotation.from_rotvec(v) for v in rng.normal(0, 1.2, (40, 3))]
    rots += [Rotation.from_rotvec(np.pi * v / np.linalg.norm(v)) for v in rng.normal(0, 1, (6, 3))]          # half turns
    rots += [Rotation.identity(), Rotation.from_rotvec([1e-4, -2e-4, 3e-4])]
    inp = "\n".join(" ".join("%.17g" % v for v in r.as_matrix().reshape(-1)) for r in rots)
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout
    vals = np.array([[float(v) for v in ln.split()] for ln in out.strip().splitlines()])
    for r, v in zip(rots, vals):
        R = r.as_matrix()
        q = v[[1, 2, 3, 0]]                                             # (w, x, y, z) -> scipy's (x, y, z, w)
        assert np.abs(Rotation.from_quat(q).as_matrix() - R).max() < 1e-9
        assert np.abs(v[7:16].reshape(3, 3) - R).max() < 1e-9           # SetQuaternionRotation(GetQuaternionRotation)
        ang = np.linalg.norm(r.as_rotvec())
        if ang > 0.2 and ang < np.pi - 0.05:                            # (the reference zeroes small rotations and special-cases pi)
            assert np.abs(Rotation.from_rotvec(v[4:7]).as_matrix() - R).max() < 1e-9
            assert np.abs(v[16:25].reshape(3, 3) - R).max() < 1e-9
        elif ang <= 0.2 and np.abs(R - R.T).max() < 0.01 and np.trace(R) > 2.8:
            assert np.all(v[4:7] == 0)
        assert v[25] < 1e-24 and abs(v[27] - 1.0) < 1e-6
        assert v[26] < 1e-3                                             # (within 0.01 of a half turn the reference's formula is approximate)
        assert abs(v[28] - np.float32(R[1, 2])) < 1e-7                  # SetCameraT: double -> float copy


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    capi = pkg("capi")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        capi.Engine()


@pytest.mark.parametrize("env", [{"MI_BA_GPUS": "3"}, {"MI_BA_GPUS": "2", "MI_BA_TRANSPORT": "host"}])
def test_optimizer_all_ranks_without_a_gpu_end_loudly_and_together(tmp_path, env):
    """MI_BA_GPUS=N forks N - 1 ranks; when a rank cannot open its device every rank gives up (no rank waits in a collective
    the others never reach) and the file is left alone"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sc = vsfm_scene(m=8, n=200, d=4, seed=3)
    d = tmp_path / "c"; d.mkdir()
    nvm = str(d / "outputVSFM_GB.nvm")
    write_nvm(nvm, sc)
    before = open(nvm).read()
    exe = os.path.join(ROOT, "large-scale-sfm_amd", "bin", "optimizer_all")
    out = subprocess.run([exe, str(d) + "/", "0", "20", "16", "32", "0.5", "0"], capture_output=True, text=True, timeout=120,
                         env={**os.environ, **env})
    assert out.returncode == 1 and "optimizer_all:" in out.stderr
    assert open(nvm).read() == before
    bad = subprocess.run([exe, str(d) + "/"], capture_output=True, text=True, timeout=60, env={**os.environ, "MI_BA_GPUS": "65"})
    assert bad.returncode == 2 and "MI_BA_GPUS" in bad.stderr


def test_side_files_are_read_as_the_callers_write_them(tmp_path):
    """the input half of the optimizer_all contract: RTglobalmapped (R + CENTRE), focal, Invmap, ours_new, Distortion written with
    the reference's printf formats (TriangulationMultiStep.cpp:1031-1088), no NVM anywhere; t = -R C, names through Invmap"""
    L = model_lib()
    sc = vsfm_scene(m=9, n=150, d=4, seed=6)
    d = tmp_path / "cluster3"; d.mkdir()
    w = write_side_files(str(d), sc, drop_points=(7, 8))
    h = L.mi_model_load_side_files((str(d) + "/").encode())
    assert h, L.mi_model_last_error()
    mdl = read_model(L, h)
    assert mdl["f"].shape == (9,) and mdl["xyz"].shape == (148, 3)
    assert np.abs(mdl["R"].reshape(9, 3, 3) - w["R"]).max() < 1e-9 and np.abs(mdl["f"] - w["f"]).max() < 1e-8
    assert np.abs(mdl["t"] - w["t"]).max() < 1e-7                   # t = -R C from 9-decimal text
    assert np.abs(mdl["xyz"] - w["X"][w["kept"]]).max() < 1e-6      # %lf: six decimals
    keep_obs = np.isin(sc.obs_pt(), w["kept"])
    assert np.array_equal(mdl["cam"], sc.obs_cam[keep_obs]) and np.abs(mdl["xy"] - sc.x[keep_obs]).max() < 1e-6
    assert np.array_equal(mdl["fid"], 5000 + 7 * sc.obs_pt()[keep_obs]) and (mdl["rgb"] == 255).all() and (mdl["d"] == 0).all()
    assert np.array_equal(mdl["pt"], np.searchsorted(w["kept"], sc.obs_pt()[keep_obs]))
    out = str(tmp_path / "o.nvm")
    assert L.mi_model_save_nvm(h, out.encode()) == 0
    names = [l.split()[0] for l in open(out).read().splitlines()[2:11]]
    assert names == [str(v) for v in w["names"]]                    # camera names of the output NVM = Invmap's
    L.mi_model_free(h)
    # Invmap and Distortion are optional (MergeClusters.cpp:1009-1032 writes no Distortion.txt): names then come from the focal file
    d2 = tmp_path / "cluster4"; d2.mkdir()
    w2 = write_side_files(str(d2), sc, invmap=False, distortion=False, cam_names=range(40, 49))
    h = L.mi_model_load_side_files(str(d2).encode())
    assert h, L.mi_model_last_error()
    assert L.mi_model_save_nvm(h, out.encode()) == 0
    assert [l.split()[0] for l in open(out).read().splitlines()[2:11]] == [str(v) for v in range(40, 49)]
    L.mi_model_free(h)
    # failures are loud
    os.remove(d2 / "ours_new.txt")
    assert not L.mi_model_load_side_files(str(d2).encode()) and b"ours_new.txt" in L.mi_model_last_error()
    with open(d / "ours_new.txt", "a") as fo:
        fo.write("0.1 0.2 0.3 255 255 255 2 0 1 0.5 0.5 99 1 0.25 0.25 \n")
    assert not L.mi_model_load_side_files(str(d).encode()) and b"camera index out of range" in L.mi_model_last_error()


def test_nvm_round_trip(tmp_path):
    L = model_lib()
    sc = vsfm_scene(m=8, n=120, d=4, seed=4)
    a = str(tmp_path / "a.nvm"); b = str(tmp_path / "b.nvm")
    write_nvm(a, sc)
    h = L.mi_model_load_nvm(a.encode())
    assert h, L.mi_model_last_error()
    mdl = read_model(L, h)
    assert mdl["f"].shape == (8,) and mdl["xyz"].shape == (120, 3) and len(mdl["pt"]) == sc.nobs
    assert np.allclose(mdl["xy"], sc.x, rtol=1e-11) and np.array_equal(mdl["cam"], sc.obs_cam)
    assert np.array_equal(mdl["pt"], sc.obs_pt()) and np.array_equal(mdl["fid"], 1000 + sc.obs_pt())
    assert np.allclose(mdl["R"], sc.R0, atol=1e-11) and np.allclose(mdl["t"], sc.p0[:8 * 7].reshape(8, 7)[:, :3], rtol=1e-11)
    # host-side RMSE of the loaded model = the oracle's cost function
    from oracle import Oracle
    assert abs(L.mi_model_rmse(h) - np.sqrt(Oracle().cost(sc, sc.p0) / sc.nobs)) < 1e-8
    assert L.mi_model_save_nvm(h, b.encode()) == 0
    h2 = L.mi_model_load_nvm(b.encode())
    m2 = read_model(L, h2)
    for k in mdl:
        assert np.allclose(mdl[k], m2[k], rtol=1e-11, atol=1e-11), k
    L.mi_model_free(h); L.mi_model_free(h2)
    assert not L.mi_model_load_nvm(str(tmp_path / "missing.nvm").encode())


def test_nvm_quaternion_variant_and_number_spellings(tmp_path):
    """the older NVM_V3 header: rotation as a quaternion, then the camera CENTRE (util_VSFM.h:176-181,
    DataInterface.h:183-207, 279-283); and the spellings a number can take in a hand-edited file"""
    from helpers import rodrigues
    L = model_lib()
    sc = vsfm_scene(m=6, n=40, d=3, seed=8)
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = sc.p0[:m * cnp].reshape(m, cnp); X = sc.p0[m * cnp:].reshape(n, 3)
    R = np.einsum("mij,mjk->mik", rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3)); t = a[:, :3]
    path = str(tmp_path / "q.nvm")
    with open(path, "w") as fo:
        fo.write("NVM_V3\n\n%d\n" % m)
        for j in range(m):
            Rj = R[j]
            qw = 0.5 * np.sqrt(max(1e-30, 1 + Rj[0, 0] + Rj[1, 1] + Rj[2, 2]))
            q = [qw, (Rj[2, 1] - Rj[1, 2]) / (4 * qw), (Rj[0, 2] - Rj[2, 0]) / (4 * qw), (Rj[1, 0] - Rj[0, 1]) / (4 * qw)]
            C = -Rj.T @ t[j]
            fo.write("img%d.jpg\t+%.10e " % (j, a[j, 6] / 0.001) + " ".join("%+.15e" % v for v in q + list(C)) + " 0 0\r\n")
        fo.write("\n%d\n" % n)
        for i in range(n):
            o0, o1 = sc.pt_ptr[i], sc.pt_ptr[i + 1]
            fo.write("%.9E %.9g %.9g 1 2 3 %d " % (*X[i], o1 - o0))
            fo.write(" ".join("%d %d %.6f %.6f" % (sc.obs_cam[o], 7 * o, sc.x[o, 0], sc.x[o, 1]) for o in range(o0, o1)) + "\n")
        fo.write("\n0\n\n#the last part of NVM file points to the PLY files\n0\n")
    h = L.mi_model_load_nvm(path.encode())
    assert h, L.mi_model_last_error()
    mdl = read_model(L, h)
    assert np.abs(mdl["R"].reshape(m, 3, 3) - R).max() < 1e-12 and np.abs(mdl["t"] - t).max() < 1e-10
    assert np.abs(mdl["f"] - a[:, 6] / 0.001).max() < 1e-6 and np.abs(mdl["xyz"] - X).max() < 1e-8
    assert np.array_equal(mdl["cam"], sc.obs_cam) and np.abs(mdl["xy"] - sc.x).max() < 1e-6 and mdl["fid"][5] == 35
    L.mi_model_free(h)
    open(path, "w").write("NVM_V3_R9T\n3\nimg 800 1 0 0\n")                       # truncated in the first camera
    assert not L.mi_model_load_nvm(path.encode()) and b"NVM" in L.mi_model_last_error()


def test_bal_round_trip_and_convention(tmp_path):
    """BAL text is the -z (Bundler) camera; the loader flips it to +z exactly as LoadBundlerModel does"""
    L = model_lib()
    sc = pkg("scene").make_scene("x", m=7, n=90, d=3, seed=6, explicit_centers=0)      # Bundler frame, t-form
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = sc.p0[:m * cnp].reshape(m, cnp); X = sc.p0[m * cnp:].reshape(n, 3)
    w = pkg("scene").log_so3(sc.R0.reshape(m, 3, 3))
    path = str(tmp_path / "p.txt")
    order = np.random.default_rng(0).permutation(sc.nobs)                              # BAL need not be sorted by point
    with open(path, "w") as fo:
        fo.write("%d %d %d\n" % (m, n, sc.nobs))
        pt = sc.obs_pt()
        for o in order:
            fo.write("%d %d %.12g %.12g\n" % (sc.obs_cam[o], pt[o], sc.x[o, 0], sc.x[o, 1]))
        for j in range(m):
            fo.write(" ".join("%.14g" % v for v in (*w[j], *a[j, :3], a[j, 6] / 0.001, 0.0, 0.0)) + "\n")
        for i in range(n):
            fo.write("%.12g %.12g %.12g\n" % tuple(X[i]))
    h = L.mi_model_load_bal(path.encode())
    assert h, L.mi_model_last_error()
    mdl = read_model(L, h)
    assert np.all(np.diff(mdl["pt"]) >= 0)                                             # regrouped by point
    F = np.diag([1.0, -1.0, -1.0])
    assert np.allclose(mdl["R"].reshape(m, 3, 3), F @ sc.R0.reshape(m, 3, 3), atol=1e-9)
    assert np.allclose(mdl["t"], a[:, :3] @ F, atol=1e-12)
    # same reprojection error in both conventions
    from oracle import Oracle
    assert abs(L.mi_model_rmse(h) - np.sqrt(Oracle().cost(sc, sc.p0) / sc.nobs)) < 1e-7
    out = str(tmp_path / "q.txt")
    assert L.mi_model_save_bal(h, out.encode()) == 0
    h2 = L.mi_model_load_bal(out.encode())
    m2 = read_model(L, h2)
    for k in ("f", "R", "t", "xyz", "xy", "pt", "cam"):
        assert np.allclose(mdl[k], m2[k], rtol=1e-9, atol=1e-9), k
    L.mi_model_free(h); L.mi_model_free(h2)


def test_bundler_out_round_trip_and_invisible_points(tmp_path):
    L = model_lib()
    sc = vsfm_scene(m=30, n=900, d=6, seed=9)
    nvm = str(tmp_path / "a.nvm"); out = str(tmp_path / "bundle.out")
    write_nvm(nvm, sc)
    h = L.mi_model_load_nvm(nvm.encode())
    assert L.mi_model_save_bundler_out(h, out.encode()) == 0
    assert os.path.exists(str(tmp_path / "bundle-list.txt"))
    h2 = L.mi_model_load_bundler_out(out.encode())
    assert h2, L.mi_model_last_error()
    a, b = read_model(L, h), read_model(L, h2)
    for k in a:
        assert np.allclose(a[k], b[k], rtol=1e-10, atol=1e-10), k
    assert abs(L.mi_model_rmse(h) - L.mi_model_rmse(h2)) < 1e-9
    assert open(out).readline().startswith("# Bundle file v0.3")
    assert L.mi_model_remove_invisible_points(h) == 0                 # every point is in front of its cameras
    L.mi_model_free(h); L.mi_model_free(h2)
    # push 25 points behind their cameras: they, and nothing else, go (every camera keeps > 20 points)
    bad = sc.p0.copy()
    X = bad[sc.m * sc.cnp:].reshape(sc.n, 3)
    X[:25] *= -40.0
    write_nvm(nvm, sc, bad)
    h = L.mi_model_load_nvm(nvm.encode())
    before = read_model(L, h)
    z = np.einsum("kj,kj->k", before["R"].reshape(-1, 3, 3)[before["cam"]][:, 2, :], before["xyz"][before["pt"]]) + before["t"][before["cam"], 2]
    gone = np.unique(before["pt"][z <= np.sort(z)[len(z) // 2] * 0.001])       # util_VSFM.h:739-761
    assert 15 <= len(gone) <= 25 and gone.max() < 25
    assert L.mi_model_remove_invisible_points(h) == len(gone)
    c = read_model(L, h)
    kept = np.setdiff1d(np.arange(sc.n), gone)
    assert c["xyz"].shape[0] == len(kept) and len(c["pt"]) == sc.nobs - 6 * len(gone) and c["f"].shape[0] == sc.m
    assert np.array_equal(c["rgb"][:, 0], kept % 256)                          # colours follow their points
    assert np.all(np.diff(c["pt"]) >= 0) and c["pt"].max() == len(kept) - 1
    L.mi_model_free(h)


def _check_plan(sc, rank=0, world=1, tmb=0, all_rows=False):
    capi = pkg("capi")
    pl = capi.Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam, rank, world, tmb, all_rows=all_rows)
    st = pl.stats()
    g = {k: pl.get(k) for k in ("perm", "obs_cam", "obs_pt", "obs_src", "pt_obs", "batch_obs", "batch_pt", "batch_pair", "tile_batch",
                                "tile_slot", "slot_gblk", "tile_cslot", "cslot_cam", "pair", "obs_cslot", "up_row",
                                "up_col", "bsr_row", "bsr_col", "bsr_src", "bsr_trans", "tile_mode", "obs_rec", "rec_cam", "wk_blk", "wk_q",
                                "rpair", "wk_diag_cam", "wk_diag_blk", "long_pts", "sb_blk", "sb_q", "sb_ent", "sb8_blk", "sb8_q", "sb8_ent")}
    assert sorted(g["perm"]) == list(range(sc.n))
    # observations: a permutation of this shard's original observations, cameras ascending inside a point
    assert np.array_equal(sc.obs_cam[g["obs_src"]], g["obs_cam"])
    pts_local = g["perm"][st["pt_begin"]:st["pt_end"]]
    assert np.array_equal(sc.obs_pt()[g["obs_src"]], pts_local[g["obs_pt"]])
    # upper pattern lookup
    up = {}
    for j in range(sc.m):
        for q in range(g["up_row"][j], g["up_row"][j + 1]):
            up[(j, g["up_col"][q])] = q
    # full pattern is symmetric and points at the right upper block
    for j in range(sc.m):
        cols = g["bsr_col"][g["bsr_row"][j]:g["bsr_row"][j + 1]]
        assert np.all(np.diff(cols) > 0)
        for q in range(g["bsr_row"][j], g["bsr_row"][j + 1]):
            k = g["bsr_col"][q]
            assert g["bsr_src"][q] == up[(min(j, k), max(j, k))] and g["bsr_trans"][q] == (k < j)
    # pairs: every (a <= b) pair of every point exactly once, slot / gblk = the (cam_a, cam_b) block
    nb, seen = len(g["batch_obs"]) - 1, 0
    tile_of_batch = np.repeat(np.arange(len(g["tile_batch"]) - 1), np.diff(g["tile_batch"]))
    row_pts = set()
    for b in range(nb):
        o0, o1 = g["batch_obs"][b], g["batch_obs"][b + 1]
        assert 0 < o1 - o0 <= 64 and g["batch_pt"][b + 1] - g["batch_pt"][b] <= 32
        t = tile_of_batch[b]
        lds = g["tile_mode"][t] in (1, 2, 3, 4)
        if g["tile_mode"][t] in (3, 4):  # one camera set (k_build_uni; k_build_mfma): lane = point * d + camera slot, slot of (a, b) = its rank in the a <= b enumeration
            d = g["tile_cslot"][t + 1] - g["tile_cslot"][t]
            assert 4 <= d <= 10 and g["tile_slot"][t + 1] - g["tile_slot"][t] == d * (d + 1) // 2 and (o1 - o0) % d == 0
            cams = g["cslot_cam"][g["tile_cslot"][t]:g["tile_cslot"][t + 1]]
            assert np.all(np.diff(cams) > 0)
            assert np.array_equal(g["obs_cam"][o0:o1], np.tile(cams, (o1 - o0) // d)) and np.array_equal(g["obs_cslot"][o0:o1], np.tile(np.arange(d), (o1 - o0) // d))
            assert np.array_equal(g["obs_pt"][o0:o1] - g["batch_pt"][b], np.repeat(np.arange((o1 - o0) // d), d))
            want_blk = [up[(cams[a], cams[c])] for a in range(d) for c in range(a, d)]
            assert list(g["slot_gblk"][g["tile_slot"][t]:g["tile_slot"][t + 1]]) == want_blk
        off, nr = int(g["batch_pair"][b]) & ((1 << 48) - 1), int(g["batch_pair"][b]) >> 48      # batches of one camera set share their entries
        ents = g["pair"][off:off + 64 * nr]
        if not lds:                 # a row tile: no pair entries, every observation has a record
            assert len(ents) == 0 and np.all(g["obs_rec"][o0:o1] >= 0)
            row_pts |= set(range(g["batch_pt"][b], g["batch_pt"][b + 1]))
            seen += o1 - o0
            continue
        assert len(ents) % 64 == 0 and np.all(g["obs_rec"][o0:o1] == -1)
        got = set()
        for q, e in enumerate(ents):
            if e == 0xFFFFFFFF:
                assert lds
                continue
            la, lb = int(e & 63), int((e >> 6) & 63)
            assert la <= lb < o1 - o0 and g["obs_pt"][o0 + la] == g["obs_pt"][o0 + lb]
            ja, jb = g["obs_cam"][o0 + la], g["obs_cam"][o0 + lb]
            blk = g["slot_gblk"][g["tile_slot"][t] + ((e >> 12) & 63)]
            assert ((e >> 12) & 63) < g["tile_slot"][t + 1] - g["tile_slot"][t] <= 55
            assert g["obs_pt"][o0 + la] - g["batch_pt"][b] == (e >> 18) & 31
            assert g["cslot_cam"][g["tile_cslot"][t] + g["obs_cslot"][o0 + la]] == ja
            assert blk == up[(ja, jb)]
            got.add((la, lb))
        want = set()
        for p in range(g["batch_pt"][b], g["batch_pt"][b + 1]):
            s, e_ = g["pt_obs"][p] - o0, g["pt_obs"][p + 1] - o0
            want |= {(a, c) for a in range(s, e_) for c in range(a, e_)}
        assert got == want
        seen += o1 - o0
    assert seen + sum(np.diff(g["pt_obs"])[g["long_pts"]]) == st["nobs_local"]
    # ---- the row path: records camera-major with points ascending inside a camera; every (a < b) pair of every row point
    # exactly once in the list of its block, in point order; a diagonal item for every camera that has records
    row_pts |= set(int(p) for p in g["long_pts"])
    assert len(row_pts) == st["nrow_pts"]
    rec_of = g["obs_rec"]
    have = np.flatnonzero(rec_of >= 0)
    assert len(have) == st["nrec"] == g["rec_cam"][-1] and sorted(rec_of[have]) == list(range(len(have)))
    rec_obs = np.zeros(len(have), np.int64); rec_obs[rec_of[have]] = have
    for j in range(sc.m):
        r = rec_obs[g["rec_cam"][j]:g["rec_cam"][j + 1]]
        assert np.all(g["obs_cam"][r] == j) and np.all(np.diff(g["obs_pt"][r]) > 0)
    assert list(g["wk_diag_cam"]) == [j for j in range(sc.m) if g["rec_cam"][j + 1] > g["rec_cam"][j]]
    assert all(g["wk_diag_blk"][k] == up[(j, j)] for k, j in enumerate(g["wk_diag_cam"]))
    got = {}
    assert len(g["wk_q"]) == len(g["wk_blk"]) + 1 and g["wk_q"][-1] * 2 == len(g["rpair"]) and np.all(np.diff(g["wk_q"]) > 0)
    assert np.all(np.diff(g["wk_blk"]) > 0)
    for w, blk in enumerate(g["wk_blk"]):
        pr = g["rpair"][2 * g["wk_q"][w]:2 * g["wk_q"][w + 1]].reshape(-1, 2)
        oa, ob = rec_obs[pr[:, 0]], rec_obs[pr[:, 1]]
        assert np.all(g["obs_pt"][oa] == g["obs_pt"][ob]) and np.all(np.diff(g["obs_pt"][oa]) > 0)
        assert all(up[(a, b)] == blk for a, b in set(zip(g["obs_cam"][oa], g["obs_cam"][ob])))
        got[blk] = set(g["obs_pt"][oa])
    # ... or, cameras in different groups of four (eight), in the entry list of its super-block (k_schur_super): per point both groups see ONE entry
    # with the point's records of the row group's cameras (slot = camera % GS) and of the column group's; the lists of four padded to whole
    # rounds of four entries.  A point uses either the groups of four alone, or the groups of eight and, inside one, its two halves of four.
    npairs_super = 0
    for GS, kb, kq, ke in ((4, "sb_blk", "sb_q", "sb_ent"), (8, "sb8_blk", "sb8_q", "sb8_ent")):
        nsup, rnd = len(g[kq]) - 1, 4 if GS == 4 else 1
        assert len(g[kb]) == GS * GS * nsup and len(g[ke]) == 2 * GS * g[kq][-1] and np.all(np.diff(g[kq]) % rnd == 0)
        assert np.all(np.diff(np.diff(g[kq])) <= 0)                      # the longest lists first (padded lengths)
        for w in range(nsup):
            blks = g[kb][GS * GS * w:GS * GS * (w + 1)].reshape(GS, GS)
            ent = g[ke][2 * GS * g[kq][w]:2 * GS * g[kq][w + 1]].reshape(-1, 2 * GS)
            real = ent[(ent >= 0).any(axis=1)]
            assert len(real) > len(ent) - rnd and np.all(ent[len(real):] == -1)
            pts_seen = []
            for e in real:
                ra, rb = e[:GS], e[GS:]
                assert (ra >= 0).any() and (rb >= 0).any()
                oa, ob = rec_obs[ra[ra >= 0]], rec_obs[rb[rb >= 0]]
                p = g["obs_pt"][oa[0]]
                assert np.all(g["obs_pt"][oa] == p) and np.all(g["obs_pt"][ob] == p)
                pts_seen.append(p)
                for x in np.flatnonzero(ra >= 0):
                    for y in np.flatnonzero(rb >= 0):
                        ca, cb = g["obs_cam"][rec_obs[ra[x]]], g["obs_cam"][rec_obs[rb[y]]]
                        assert ca % GS == x and cb % GS == y and ca // GS < cb // GS and blks[x, y] == up[(ca, cb)]
                        assert p not in got.setdefault(blks[x, y], set())
                        got[blks[x, y]].add(p)
                        npairs_super += 1
            assert np.all(np.diff(pts_seen) > 0)                         # point order: the sums have a fixed order
    nsup = len(g["sb_q"]) - 1 + len(g["sb8_q"]) - 1
    assert st["npair_rows"] == len(g["rpair"]) // 2 + npairs_super
    want = {}
    for p in row_pts:
        cams = g["obs_cam"][g["pt_obs"][p]:g["pt_obs"][p + 1]]
        for a in range(len(cams)):
            for b in range(a + 1, len(cams)):
                want.setdefault(up[(cams[a], cams[b])], set()).add(p)
    assert got == want
    if nsup:
        assert all(g["obs_cam"][rec_obs[pr0]] // 4 == g["obs_cam"][rec_obs[pr1]] // 4 for pr0, pr1 in g["rpair"].reshape(-1, 2))
    return st, g


def test_plan_tables_coherent_scene():
    sc = pkg("scene").make_scene("C1")
    st, g = _check_plan(sc)
    assert st["nbatch_rows"] == 0 and st["npts_local"] == sc.n and st["nb_upper"] == len(g["up_col"]) and st["nrec"] == 0
    st, g = _check_plan(sc, all_rows=True)        # the deterministic mode: every point on the row path
    assert st["nbatch_rows"] == st["nbatch"] and st["nrec"] == sc.nobs and st["npair"] == 0


def test_plan_tables_with_matrix_pipe_tiles(monkeypatch):
    """MI_BA_MFMA=1: tiles of one camera set become mode 3 and end where the set changes; what k_build_mfma relies on (lane =
    point * d + camera slot, slots in the a <= b enumeration) is checked by _check_plan"""
    monkeypatch.setenv("MI_BA_MFMA", "1")
    for d in (4, 7, 10):
        sc = pkg("scene").make_scene("x", m=24, n=2400, d=d, seed=100 + d, strides=(1,))
        assert (pkg("capi").Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam).get("tile_mode") == 3).mean() > 0.6
        _check_plan(sc)
        _check_plan(sc, rank=1, world=2)


def test_plan_tables_with_uniform_tiles(monkeypatch):
    """the default: tiles of one camera set are mode 4 (k_build_uni) and end where the set changes; what the kernel relies on -- lane =
    point * d + camera slot, slots in the a <= b enumeration, whole batches of floor(64 / d) points but the tile's last -- is checked
    here and by _check_plan; MI_BA_UNI=0 keeps the general kernel's tiles"""
    for d in (4, 7, 10):
        sc = pkg("scene").make_scene("x", m=24, n=2400, d=d, seed=100 + d, strides=(1,))
        pl = pkg("capi").Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam)
        modes = pl.get("tile_mode")
        assert (modes == 4).mean() > 0.6 and not (modes == 3).any()
        tb, bp = pl.get("tile_batch"), pl.get("batch_pt")
        for t in np.where(modes == 4)[0]:
            npts = np.diff(bp[tb[t]:tb[t + 1] + 1])
            assert np.all(npts[:-1] == 64 // d) and 0 < npts[-1] <= 64 // d
        _check_plan(sc)
        _check_plan(sc, rank=1, world=2)
    monkeypatch.setenv("MI_BA_UNI", "0")
    assert not (pkg("capi").Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam).get("tile_mode") == 4).any()


def test_back_substitution_batches_and_uniform_tiles_partition_the_points(monkeypatch):
    """k_backsub walks the plan's bs batches (runs of consecutive points, <= 64 observations, <= 32 points), k_backsub_uni the tiles of one
    camera set (mode 4): together they cover every short point exactly once; MI_BA_UNI_BS=0 hands every point to k_backsub"""
    scene = pkg("scene")
    rng = np.random.default_rng(5)
    L = np.minimum(2 + rng.geometric(0.3, 3000), 12)
    cases = [scene.make_scene("x", m=24, n=2400, d=7, seed=107, strides=(1,)), scene.make_scene("C1"),
             scene.make_scene("x", m=40, n=3000, d=12, seed=9, strides=(1, 2), track_len=L)]
    for sc in cases:
        for env in (None, "0"):
            if env is None:
                monkeypatch.delenv("MI_BA_UNI_BS", raising=False)
            else:
                monkeypatch.setenv("MI_BA_UNI_BS", env)
            pl = pkg("capi").Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam)
            st = pl.stats()
            b0, b1, o0, o1 = (pl.get(k) for k in ("bs_batch_pt", "bs_batch_pt_end", "bs_batch_obs", "bs_batch_obs_end"))
            pt_obs, tm, tb, bp = pl.get("pt_obs"), pl.get("tile_mode"), pl.get("tile_batch"), pl.get("batch_pt")
            nshort = st["npts_local"] - st["nlong"]
            seen = np.zeros(st["npts_local"], int)
            for k in range(len(b0)):
                assert 0 < b1[k] - b0[k] <= 32 and 0 < o1[k] - o0[k] <= 64 and o0[k] == pt_obs[b0[k]] and o1[k] == pt_obs[b1[k]]
                seen[b0[k]:b1[k]] += 1
            nuni = 0
            if env is None:
                for t in np.where(tm == 4)[0]:
                    seen[bp[tb[t]]:bp[tb[t + 1]]] += 1
                    nuni += 1
            assert np.all(seen[:nshort] == 1) and np.all(seen[nshort:] == 0)
            assert (nuni > 0) == (env is None and (tm == 4).any())


def test_plan_tables_ragged_scene():
    """track lengths 1..20 (register tiles and row tiles), tiny tile budget"""
    scene = pkg("scene")
    rng = np.random.default_rng(1)
    base = scene.make_scene("x", m=40, n=500, d=20, seed=2, strides=(1,))
    keep = [np.sort(rng.permutation(np.arange(base.pt_ptr[i], base.pt_ptr[i + 1]))[:rng.integers(1, 21)]) for i in range(base.n)]
    # a camera set has to repeat to earn a register-accumulated tile (plan.h kMinRunPts): the first 350 points keep a
    # prefix of their views whose length depends on the start camera only, so ~9 points share each camera set
    start = base.obs_cam[base.pt_ptr[:-1]]
    for i in range(350):
        keep[i] = np.arange(base.pt_ptr[i], base.pt_ptr[i] + 1 + (int(start[i]) * 7) % 20)
    sel = np.concatenate(keep)
    sc = scene.Scene(n=base.n, m=base.m, cnp=7, est_focal=1, undistort=0, explicit_centers=1, proj_sign=-1.0,
                     pt_ptr=np.concatenate([[0], np.cumsum([len(k) for k in keep])]).astype(np.int32),
                     obs_cam=base.obs_cam[sel], x=base.x[sel], R0=base.R0, f_fixed=base.f_fixed, p0=base.p0, p_true=base.p_true)
    os.environ["MI_BA_MIN_RUN_PTS"] = "6"
    try:
        st, g = _check_plan(sc, tmb=3)
    finally:
        del os.environ["MI_BA_MIN_RUN_PTS"]
    assert 0 < st["nbatch_rows"] < st["nbatch"] and 0 < st["nrow_pts"] < sc.n


@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_partition_the_points(world):
    sc = pkg("scene").make_scene("C1")
    capi = pkg("capi")
    owned, pattern = [], None
    for r in range(world):
        st, g = _check_plan(sc, r, world)
        owned.append(g["perm"][st["pt_begin"]:st["pt_end"]])
        pat = (tuple(g["up_row"]), tuple(g["up_col"]))
        assert pattern in (None, pat)              # every rank derives the same block pattern of S
        pattern = pat
    allp = np.concatenate(owned)
    assert len(allp) == sc.n and len(set(allp)) == sc.n
    sizes = [len(o) for o in owned]
    assert max(sizes) - min(sizes) <= 0.1 * sc.n / world + 8


def test_plan_rejects_bad_input():
    capi = pkg("capi")
    with pytest.raises(ValueError, match="sorted by point"):
        capi.Plan(3, 2, [0, 2, 1], [0, 0, 1])
    with pytest.raises(ValueError, match="twice"):
        capi.Plan(2, 2, [0, 0, 1], [1, 1, 0])
    with pytest.raises(ValueError, match="out of range"):
        capi.Plan(2, 2, [0, 1], [0, 5])


def test_plan_with_more_long_tracks_than_a_grid_dimension():
    """66 000 tracks of 65 views: more points beyond a wavefront than the 65 535 a grid's y dimension can index.  Round 1 launched a
    kernel with the long points on blockIdx.y; since round 2 they take one workgroup each on blockIdx.x (k_long_reduce,
    k_long_backsub) and their pairs are k_schur_*'s block-major work lists, whatever their number.  The planner's side: all of
    them long, every pair of every track in the row path's lists exactly once."""
    m, n, d = 70, 66000, 65
    start = (np.arange(n) * 7) % m
    cams = np.sort((start[:, None] + np.arange(d)[None, :]) % m, axis=1).astype(np.int32).reshape(-1)
    pts = np.repeat(np.arange(n, dtype=np.int32), d)
    capi = pkg("capi")
    pl = capi.Plan(n, m, pts, cams)
    st = pl.stats()
    assert st["nlong"] == n > 65535 and st["ntile"] == 0
    assert st["nrec"] == n * d and st["npair_rows"] == n * (d * (d + 1) // 2 - d)      # off-diagonal pairs; the diagonal ones are k_schur_blocks<DIAG>'s
    long_pts = pl.get("long_pts")
    assert len(long_pts) == n and len(set(long_pts.tolist())) == n


def test_large_nvm_files_read_and_written_by_threads_equal_the_serial_path(tmp_path, monkeypatch):
    """models beyond 32 768 points: the point records are parsed / formatted by several host threads (model_io.cpp).  Same arrays,
    same bytes as the serial reader and writer (MI_MODEL_IO_THREADS=1); a file whose records do not end with their lines goes
    through the token-by-token reader and still loads"""
    L = model_lib()
    sc = vsfm_scene(m=12, n=40000, d=4, seed=21)
    src = str(tmp_path / "in.nvm")
    write_nvm(src, sc)
    assert os.path.getsize(src) > (1 << 20)
    out = {}
    for mode in ("threads", "serial"):
        if mode == "serial":
            monkeypatch.setenv("MI_MODEL_IO_THREADS", "1")
        h = L.mi_model_load_nvm(src.encode())
        assert h, L.mi_model_last_error()
        mdl = read_model(L, h)
        dst = str(tmp_path / (mode + ".nvm"))
        assert L.mi_model_save_nvm(h, dst.encode()) == 0
        L.mi_model_free(h)
        out[mode] = (mdl, open(dst, "rb").read())
    a, b = out["threads"], out["serial"]
    for k in a[0]:
        assert np.array_equal(np.asarray(a[0][k]), np.asarray(b[0][k])), k
    assert a[1] == b[1]
    monkeypatch.delenv("MI_MODEL_IO_THREADS")
    # records broken over two lines: the threaded reader declines, the stream reader takes over
    text = open(src).read().split("\n")
    head, pts = text[:2 + 12 + 1], text[2 + 12 + 1:]
    broken = []
    for k, line in enumerate(pts):
        toks = line.split(" ")
        broken.append(" ".join(toks[:5]) + "\n" + " ".join(toks[5:]) if (k % 3 == 0 and len(toks) > 8) else line)
    src2 = str(tmp_path / "broken.nvm")
    open(src2, "w").write("\n".join(head + broken))
    h = L.mi_model_load_nvm(src2.encode())
    assert h, L.mi_model_last_error()
    mdl2 = read_model(L, h); L.mi_model_free(h)
    for k in a[0]:
        assert np.array_equal(np.asarray(a[0][k]), np.asarray(mdl2[k])), k


def test_large_side_files_read_by_threads_equal_the_serial_path(tmp_path, monkeypatch):
    """ours_new.txt of a large cluster (the same point records as an NVM, no count in front): threaded and serial readers agree"""
    from helpers import write_side_files
    L = model_lib()
    sc = vsfm_scene(m=12, n=40000, d=4, seed=22)
    d = tmp_path / "c"; d.mkdir()
    write_side_files(str(d), sc)
    assert os.path.getsize(str(d / "ours_new.txt")) > (1 << 20)
    out = []
    for mode in ("threads", "serial"):
        if mode == "serial":
            monkeypatch.setenv("MI_MODEL_IO_THREADS", "1")
        h = L.mi_model_load_side_files((str(d) + "/").encode())
        assert h, L.mi_model_last_error()
        out.append(read_model(L, h)); L.mi_model_free(h)
    for k in out[0]:
        assert np.array_equal(np.asarray(out[0][k]), np.asarray(out[1][k])), k
    assert len(out[0]["xyz"]) == sc.n



def test_plan_tables_with_super_blocks(monkeypatch):
    """MI_BA_SUPER=4 (the default from 2^20 row-path pairs on; 8: groups of eight as well, opt-in): the pairs whose cameras lie in different groups of four
    (eight) consecutive cameras sit in super-block entry lists, the others stay in the per-block pair lists; together every pair of every row point exactly once (_check_plan)"""
    scene = pkg("scene")
    rng = np.random.default_rng(11)
    L = np.minimum(2 + rng.geometric(0.12, 1500), 40)
    for sc in (scene.make_scene("x", m=83, n=1500, d=40, seed=21, strides=(1, 2), track_len=L),
               scene.make_scene("x", m=40, n=600, d=13, seed=5, strides=(1, 2, 3))):
        monkeypatch.setenv("MI_BA_SUPER", "8")
        st, g = _check_plan(sc)
        assert len(g["sb_q"]) > 1 and len(g["sb8_q"]) > 1                         # (both scenes have tracks that reach four groups of eight)
        _check_plan(sc, rank=1, world=2)
        _check_plan(sc, all_rows=True)
        monkeypatch.setenv("MI_BA_SUPER", "4")
        st, g = _check_plan(sc)
        assert len(g["sb_q"]) > 1 and len(g["sb8_q"]) == 1
    monkeypatch.setenv("MI_BA_SUPER", "0")
    st, g = _check_plan(sc)
    assert len(g["sb_q"]) == 1 and len(g["sb_ent"]) == 0 and len(g["sb8_ent"]) == 0


def test_super_blocks_only_where_entries_hold_several_pairs(monkeypatch):
    """by default (no MI_BA_SUPER) a scene with 2^20 row-path pairs or more gets super-blocks when the cameras that see a point are neighbours in the
    numbering (13 consecutive cameras: ~10 pairs per entry) and keeps the per-block pair lists when they are not (every 31st camera: one pair per entry)"""
    monkeypatch.delenv("MI_BA_SUPER", raising=False)
    scene, capi = pkg("scene"), pkg("capi")
    for stride, want in ((1, True), (31, False)):
        sc = scene.make_scene("x", m=410, n=14000, d=13, seed=3, strides=(stride,))
        pl = capi.Plan(sc.n, sc.m, sc.obs_pt(), sc.obs_cam)
        st = pl.stats()
        assert st["npair_rows"] == sc.n * 78 >= 1 << 20
        assert (len(pl.get("sb_q")) > 1) == want and (len(pl.get("rpair")) // 2 < st["npair_rows"]) == want

