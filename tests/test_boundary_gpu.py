"""The drop-in entry points on the GPU: run_sfm / sba_motstr_levmar shaped C calls against the REAL
reference's outputs (golden), the ParallelBA C++ plugin, NVM files through the optimizer_all binary."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, pkg, scene_from_golden
from helpers import model_lib, read_model, vsfm_scene, write_nvm

pytestmark = pytest.mark.gpu


class CameraParams(C.Structure):      # sfm.h:29-51 / include/mi_sfm.h
    _fields_ = [("R", C.c_double * 9), ("t", C.c_double * 3), ("f", C.c_double), ("k", C.c_double * 2),
                ("k_inv", C.c_double * 6), ("constrained", C.c_char * 9), ("constraints", C.c_double * 9),
                ("weights", C.c_double * 9), ("K_known", C.c_double * 9), ("k_known", C.c_double * 5),
                ("fisheye", C.c_char), ("known_intrinsics", C.c_char), ("f_cx", C.c_double), ("f_cy", C.c_double),
                ("f_rad", C.c_double), ("f_angle", C.c_double), ("f_focal", C.c_double), ("f_scale", C.c_double),
                ("k_scale", C.c_double)]


def _dense_inputs(sc):
    m, n = sc.m, sc.n
    vmask = np.zeros((n, m), np.int8)
    vmask[sc.obs_pt(), sc.obs_cam] = 1
    return vmask, np.ascontiguousarray(sc.x)


@pytest.mark.parametrize("name", ["C1", "C1t"])
def test_run_sfm_matches_the_references_run_sfm(name):
    sc, z = scene_from_golden(name)
    L = pkg("capi").lib()
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = sc.p0[:m * cnp].reshape(m, cnp)
    cams = (CameraParams * m)()
    for j in range(m):
        cams[j].R[:] = list(sc.R0[j]); cams[j].t[:] = list(a[j, :3]); cams[j].f = a[j, 6] / 0.001
    pts = np.ascontiguousarray(sc.p0[m * cnp:]).copy()
    vmask, x = _dense_inputs(sc)
    info = np.zeros(10)
    L.mi_run_sfm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_double,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    ret = L.mi_run_sfm(n, m, 0, vmask.ctypes.data, x.ctypes.data, 1, 0, 0, sc.explicit_centers, C.addressof(cams),
                       pts.ctypes.data, 0, 0, None, 0.0, 0, 0, 1e-12, None, None, None, None, info.ctypes.data)
    assert ret == int(z["ret_final"]) and info[6] == z["info_final"][6]
    R = np.array([list(c.R) for c in cams]).reshape(m, 3, 3)
    t = np.array([list(c.t) for c in cams]); f = np.array([c.f for c in cams])
    assert np.abs(R - z["sfm_R"]).max() < 1e-4 and np.abs(t - z["sfm_t"]).max() < 5e-3
    assert np.abs(f - z["sfm_f"]).max() < 0.5 and np.abs(pts.reshape(n, 3) - z["sfm_pts"]).max() < 5e-3
    rm, rw = np.sqrt(info[1] / sc.nobs), np.sqrt(z["info_final"][1] / sc.nobs)
    assert abs(rm - rw) <= max(1e-3, 0.005 * rw)
    # unsupported options are refused, not silently ignored
    assert L.mi_run_sfm(n, m, 0, vmask.ctypes.data, x.ctypes.data, 1, 0, 1, 1, C.addressof(cams), pts.ctypes.data,
                        0, 0, None, 0.0, 0, 0, 1e-12, None, None, None, None, None) == -1


def test_sba_motstr_levmar_shape():
    sc, z = scene_from_golden("C1")
    L = pkg("capi").lib()
    vmask, x = _dense_inputs(sc)
    p = sc.p0.copy(); info = np.zeros(10)
    opts = np.array([1e-3, 1e-10, 1e-12, 1e-12, 0.0, 4e-2])
    L.mi_sba_motstr_levmar.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ret = L.mi_sba_motstr_levmar(sc.n, sc.m, 2, vmask.ctypes.data, p.ctypes.data, sc.cnp, x.ctypes.data, sc.R0.ctypes.data,
                                 sc.f_fixed.ctypes.data, 1, 150, 0, opts.ctypes.data, info.ctypes.data)
    want = z["info_mcon2"]
    assert ret == int(want[5]) and info[6] == want[6] and info[9] == want[9]
    assert np.array_equal(p[:2 * sc.cnp], sc.p0[:2 * sc.cnp])
    assert np.abs(p - z["p_mcon2"]).max() < 1e-4            # two fixed cameras pin the gauge: parameters comparable
    assert abs(info[1] - want[1]) < 1e-6 * want[1]


def test_model_optimize_and_optimizer_all_binary(tmp_path, oracle):
    L = model_lib()
    sc = vsfm_scene(m=16, n=1500, d=5, seed=11)
    d = tmp_path / "cluster7"; d.mkdir()
    nvm = str(d / "outputVSFM_GB.nvm")
    write_nvm(nvm, sc)
    o = oracle.run(sc, itmax=50, fd_jacobian=0)
    want = np.sqrt(o["info"][1] / sc.nobs)
    # library call
    h = L.mi_model_load_nvm(nvm.encode())
    info = np.zeros(10)
    cfg = pkg("capi").default_config(itmax=50)
    it = L.mi_model_optimize(h, C.byref(cfg), info.ctypes.data_as(C.POINTER(C.c_double)))
    assert it == o["ret"] and abs(L.mi_model_rmse(h) - want) < 1e-5
    L.mi_model_free(h)
    # the process boundary, argument tuple of MergeClusters.cpp:1104
    exe = os.path.join(ROOT, "large-scale-sfm_amd", "bin", "optimizer_all")
    out = subprocess.run([exe, str(d) + "/", "0", "50", "16", "32", "0.5", "0"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "LM iterations" in out.stdout
    h = L.mi_model_load_nvm(nvm.encode())                 # the file was rewritten in place
    mdl = read_model(L, h)
    assert abs(L.mi_model_rmse(h) - want) < 1e-4          # text round trip at 12 significant digits
    assert np.array_equal(mdl["fid"], 1000 + sc.obs_pt()) and np.array_equal(mdl["rgb"][:, 0], np.arange(sc.n) % 256)
    L.mi_model_free(h)
    bad = subprocess.run([exe, str(tmp_path / "nothing") + "/", "0", "5", "16", "32", "0.5", "0"], capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


def test_parallelba_plugin_client(tmp_path, oracle):
    sc = vsfm_scene(m=16, n=1500, d=5, seed=12)
    nvm = str(tmp_path / "in.nvm"); res = str(tmp_path / "out.txt")
    write_nvm(nvm, sc)
    exe = str(tmp_path / "pba_client")
    libdir = os.path.join(ROOT, "large-scale-sfm_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "pba_client.cpp"),
                           "-o", exe, "-L", libdir, "-lmi_ba", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, nvm, res, "2"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    line = out.stdout.splitlines()[0].split()
    iters, mse = int(line[1]), float(line[3])
    # float measurements / float results: compare at float precision with the double oracle (2 fixed cameras)
    o = oracle.run(sc, itmax=150, fd_jacobian=0, mcon=2)
    assert abs(iters - o["ret"]) <= 1        # successful iterations; the closing iteration (stop 8) is not one
    assert abs(mse - o["info"][1] / sc.nobs) < 1e-3 * o["info"][1] / sc.nobs
    assert "empty run 0 version 105" in out.stdout
    vals = np.loadtxt(res, max_rows=sc.m)
    a = o["p"][:sc.m * sc.cnp].reshape(sc.m, sc.cnp)
    assert np.abs(vals[:, 0] - a[:, 6] / 0.001).max() < 0.05 and np.abs(vals[:, 1:4] - a[:, :3]).max() < 1e-3
    assert np.allclose(vals[:2, 1:4], sc.p0[:2 * sc.cnp].reshape(2, sc.cnp)[:, :3], atol=1e-5)   # constant cameras untouched
