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
    """BASELINE config C4 shape: many cameras along a closed path, every point seen by a run of neighbouring
    cameras whose length is heavy-tailed (2 .. dmax), so the reduced camera system is a wide band"""
    scene = pkg("scene")
    base = scene.make_scene("x", m=m, n=n, d=dmax, seed=seed, strides=(1,), **kw)
    rng = np.random.default_rng(seed + 1)
    L = np.minimum(2 + rng.geometric(0.14, n), dmax)
    L[rng.random(n) < 0.01] = dmax                       # a few very long tracks (> one wavefront)
    keep = np.concatenate([np.arange(base.pt_ptr[i], base.pt_ptr[i] + L[i]) for i in range(n)])
    return scene.Scene(n=n, m=m, cnp=base.cnp, est_focal=base.est_focal, undistort=0, explicit_centers=base.explicit_centers,
                       proj_sign=base.proj_sign, pt_ptr=np.concatenate([[0], np.cumsum(L)]).astype(np.int32),
                       obs_cam=base.obs_cam[keep], x=base.x[keep], R0=base.R0, f_fixed=base.f_fixed, p0=base.p0,
                       p_true=base.p_true, name="ladybug-like")


def write_bal(path, sc, p=None, shuffle_seed=None):
    """BAL / "bundler model" text (util_VSFM.h:448-501) of a Bundler-frame scene holding t (explicit_centers=0)"""
    assert sc.proj_sign == -1.0 and sc.explicit_centers == 0 and sc.est_focal
    p = sc.p0 if p is None else p
    m, n, cnp = sc.m, sc.n, sc.cnp
    a = p[:m * cnp].reshape(m, cnp); X = p[m * cnp:].reshape(n, 3)
    R = np.einsum("mij,mjk->mik", rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3))
    w = pkg("scene").log_so3(R)
    order = np.arange(sc.nobs) if shuffle_seed is None else np.random.default_rng(shuffle_seed).permutation(sc.nobs)
    pt = sc.obs_pt()
    with open(path, "w") as fo:
        fo.write("%d %d %d\n" % (m, n, sc.nobs))
        np.savetxt(fo, np.column_stack([sc.obs_cam[order], pt[order], sc.x[order]]), fmt="%d %d %.10g %.10g")
        np.savetxt(fo, np.column_stack([w, a[:, :3], a[:, 6] / 0.001, np.zeros((m, 2))]), fmt="%.14g")
        np.savetxt(fo, X, fmt="%.12g")
