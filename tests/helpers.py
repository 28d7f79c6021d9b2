"""test helpers: NVM / BAL writers (python side), scene variants."""
import numpy as np

from conftest import pkg


def rodrigues(w):
    return pkg("scene")._rodrigues(np.asarray(w, float))


def vsfm_scene(name="x", **kw):
    """scene in the file convention of NVM: +z camera, camera params hold t (x^ = f (RX+t).xy / z)"""
    return pkg("scene").make_scene(name, proj_sign=1.0, explicit_centers=0, **kw)


def write_nvm(path, sc, p=None, names=None):
    p = sc.p0 if p is None else p
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = p[:m * cnp].reshape(m, cnp)
    X = p[m * cnp:].reshape(n, 3)
    R = np.einsum("mij,mjk->mik", rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3))
    f = a[:, 6] / 0.001 if sc.est_focal else sc.f_fixed
    with open(path, "w") as fo:
        fo.write("NVM_V3_R9T\n%d\n" % m)
        for j in range(m):
            fo.write("%s %.12g " % (names[j] if names else "cam%04d.jpg" % j, f[j]))
            fo.write(" ".join("%.12g" % v for v in R[j].reshape(-1)) + " ")
            fo.write("%.12g %.12g %.12g 0 0\n" % tuple(a[j, :3]))
        fo.write("%d\n" % n)
        for i in range(n):
            o0, o1 = sc.pt_ptr[i], sc.pt_ptr[i + 1]
            fo.write("%.12g %.12g %.12g %d %d %d %d " % (*X[i], i % 256, (2 * i) % 256, (3 * i) % 256, o1 - o0))
            for o in range(o0, o1):
                fo.write("%d %d %.12g %.12g " % (sc.obs_cam[o], 1000 + i, sc.x[o, 0], sc.x[o, 1]))
            fo.write("\n")


def write_side_files(d, sc, p=None, cam_names=None, distortion=True, invmap=True, drop_points=()):
    """the five files a stage program leaves before system("../bin/optimizer_all ..."), with the reference's own printf formats
    (SiftGPU/TriangulationMultiStep.cpp:1031-1088): R and the camera CENTRE with %.9lf, "camName focal 2", "index camName",
    "0.000" per camera, and per point "%lf %lf %lf 255 255 255 n " + "%d %lld %lf %lf " per view.  No NVM is written."""
    import os
    p = sc.p0 if p is None else p
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = p[:m * cnp].reshape(m, cnp)
    X = p[m * cnp:].reshape(n, 3)
    R = np.einsum("mij,mjk->mik", rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3))
    Cc = -np.einsum("mji,mj->mi", R, a[:, :3])                      # C = -R^T t
    f = a[:, 6] / 0.001 if sc.est_focal else sc.f_fixed
    cam_names = list(cam_names) if cam_names is not None else [100 + 3 * j for j in range(m)]
    with open(os.path.join(d, "RTglobalmapped_for_optimisation.txt"), "w") as fr, \
            open(os.path.join(d, "focal_for_optimisation.txt"), "w") as ff:
        ff.write("%d\n" % m)
        for j in range(m):
            fr.write(" ".join("%.9f" % v for v in R[j].reshape(-1)) + "\n")
            fr.write("%.9f %.9f %.9f\n" % tuple(Cc[j]))
            ff.write("%d %.9f %d\n" % (cam_names[j], f[j], 2))
    if invmap:
        with open(os.path.join(d, "Invmap.txt"), "w") as fm:
            for j in range(m):
                fm.write("%d %d\n" % (j, cam_names[j]))
    if distortion:
        with open(os.path.join(d, "Distortion.txt"), "w") as fd:
            fd.write("0.000\n" * m)
    kept = []
    with open(os.path.join(d, "ours_new.txt"), "w") as fo:
        for i in range(n):
            if i in drop_points:
                continue
            o0, o1 = sc.pt_ptr[i], sc.pt_ptr[i + 1]
            fo.write("%f %f %f 255 255 255 %d " % (*X[i], o1 - o0))
            for o in range(o0, o1):
                fo.write("%d %d %f %f " % (sc.obs_cam[o], 5000 + 7 * i, sc.x[o, 0], sc.x[o, 1]))
            fo.write("\n")
            kept.append(i)
    return dict(R=R, C=Cc, t=a[:, :3].copy(), f=np.asarray(f, float), X=X.copy(), names=cam_names, kept=np.array(kept))


def read_model(mdl_lib, h):
    import ctypes as C
    cnt = (C.c_int64 * 3)()
    mdl_lib.mi_model_counts(h, cnt)
    nc, npt, npj = cnt[0], cnt[1], cnt[2]
    f = np.zeros(nc); R = np.zeros((nc, 9)); t = np.zeros((nc, 3)); d = np.zeros((nc, 2)); xyz = np.zeros((npt, 3))
    rgb = np.zeros((npt, 3), np.int32); xy = np.zeros((npj, 2)); pt = np.zeros(npj, np.int32); cam = np.zeros(npj, np.int32)
    fid = np.zeros(npj, np.int64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    mdl_lib.mi_model_get(h, P(f), P(R), P(t), P(d), P(xyz), P(rgb), P(xy), P(pt), P(cam), P(fid))
    return dict(f=f, R=R, t=t, d=d, xyz=xyz, rgb=rgb, xy=xy, pt=pt, cam=cam, fid=fid)


def model_lib():
    import ctypes as C
    L = pkg("capi").lib()
    L.mi_model_load_nvm.restype = C.c_void_p; L.mi_model_load_nvm.argtypes = [C.c_char_p]
    L.mi_model_load_bal.restype = C.c_void_p; L.mi_model_load_bal.argtypes = [C.c_char_p]
    L.mi_model_load_side_files.restype = C.c_void_p; L.mi_model_load_side_files.argtypes = [C.c_char_p]
    L.mi_model_load_bundler_out.restype = C.c_void_p; L.mi_model_load_bundler_out.argtypes = [C.c_char_p]
    L.mi_model_save_bundler_out.argtypes = [C.c_void_p, C.c_char_p]; L.mi_model_remove_invisible_points.argtypes = [C.c_void_p]
    L.mi_model_save_nvm.argtypes = [C.c_void_p, C.c_char_p]; L.mi_model_save_bal.argtypes = [C.c_void_p, C.c_char_p]
    L.mi_model_free.argtypes = [C.c_void_p]; L.mi_model_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.mi_model_get.argtypes = [C.c_void_p] * 11
    L.mi_model_last_error.restype = C.c_char_p
    L.mi_model_rmse.restype = C.c_double; L.mi_model_rmse.argtypes = [C.c_void_p]
    L.mi_model_optimize.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    return L


def ladybug_like_scene(m=1700, n=40000, dmax=120, seed=4, **kw):
    """BASELINE config C4 shape (large-scale-sfm_amd/scene.py ladybug_like_scene), at a test's size"""
    return pkg("scene").ladybug_like_scene(m=m, n=n, dmax=dmax, seed=seed, **kw)
