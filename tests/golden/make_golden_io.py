"""Golden vectors for SURVEY row a2 (model I/O): what the REFERENCE's own loaders and writers (Optimizer/util_VSFM.h, compiled
where it lies into oracle/_ref/ref_io by `make -C oracle refio`) make of small seeded model files.  Stored under
tests/golden/io/: the input files (written here, by Python), the reference's parsed arrays (text dumps, floats printed with
%.9g = exact) and the files the reference's writers produce from them.  Data only -- no reference source.

    python tests/golden/make_golden_io.py

Inputs: in_r9t.nvm (NVM_V3_R9T, distortion 0, as every stage program writes it), in_q.nvm (NVM_V3: quaternion + camera centre),
in_dist.nvm (R9T with a non-zero d0 and d1: what the reference's writers do with it is a quirk worth pinning), in.bal (BAL /
"bundler model" text, observations NOT grouped by point), in.out + in-list.txt (Bundler v0.3), in_rip.nvm (points behind cameras)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import rodrigues, vsfm_scene, write_bal, write_nvm  # noqa: E402
from conftest import pkg  # noqa: E402

OUT = os.path.join(HERE, "io")
REFIO = os.path.join(ROOT, "oracle", "_ref", "ref_io")


def ref(*args):
    subprocess.run([REFIO, *args], check=True, stdout=subprocess.DEVNULL)


def quat_of(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()          # x y z w
    return np.concatenate([q[:, 3:], q[:, :3]], 1)  # w x y z, the order DataInterface.h:183-207 reads


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refio"])
    sc = vsfm_scene(m=6, n=40, d=4, seed=31)
    m, n, cnp = sc.m, sc.n, sc.cnp
    names = ["%d" % (17 + 5 * j) for j in range(m)]
    write_nvm(os.path.join(OUT, "in_r9t.nvm"), sc, names=names)
    # NVM_V3: quaternion + camera centre
    a = sc.p0[:m * cnp].reshape(m, cnp)
    R = np.einsum("mij,mjk->mik", rodrigues(a[:, 3:6]), sc.R0.reshape(m, 3, 3))
    C = -np.einsum("mji,mj->mi", R, a[:, :3])
    q = quat_of(R)
    lines = open(os.path.join(OUT, "in_r9t.nvm")).read().splitlines()
    with open(os.path.join(OUT, "in_q.nvm"), "w") as fo:
        fo.write("NVM_V3\n%d\n" % m)
        for j in range(m):
            fo.write("%s %.12g %s %s 0 0\n" % (names[j], a[j, 6] / 0.001, " ".join("%.12g" % v for v in q[j]), " ".join("%.12g" % v for v in C[j])))
        fo.write("\n".join(lines[2 + m:]) + "\n")
    # non-zero distortion fields
    with open(os.path.join(OUT, "in_dist.nvm"), "w") as fo:
        fo.write("\n".join(lines[:2]) + "\n")
        for j in range(m):
            t = lines[2 + j].split()
            t[-2], t[-1] = "%.6g" % (-0.02 - 0.001 * j), "%.6g" % (0.003 * (j + 1))
            fo.write(" ".join(t) + "\n")
        fo.write("\n".join(lines[2 + m:]) + "\n")
    # BAL text (Bundler frame, t-form), observations shuffled
    scb = pkg("scene").make_scene("x", m=6, n=40, d=4, seed=32, explicit_centers=0)
    write_bal(os.path.join(OUT, "in.bal"), scb, shuffle_seed=5)
    # points behind their cameras, enough observations per camera for RemoveInvisiblePoints' 20-point rule
    big = vsfm_scene(m=8, n=260, d=5, seed=33)
    bad = big.p0.copy()
    bad[big.m * big.cnp:].reshape(big.n, 3)[:12] *= -40.0
    write_nvm(os.path.join(OUT, "in_rip.nvm"), big, bad, names=["c%d.jpg" % j for j in range(big.m)])

    P = lambda f: os.path.join(OUT, f)
    for stem in ("in_r9t", "in_q", "in_dist"):
        ref("load", P(stem + ".nvm"), P(stem + ".load.txt"))
        ref("loadid", P(stem + ".nvm"), P(stem + ".loadid.txt"))
        ref("save", P(stem + ".nvm"), P(stem + ".save.nvm"), "nvm")
        ref("save", P(stem + ".nvm"), P(stem + ".saveid.nvm"), "nvmid")
    ref("save", P("in_r9t.nvm"), P("in.out"), "out")            # the reference's own bundle.out (+ in-list.txt) as the .out input
    ref("load", P("in.out"), P("in.out.load.txt"))
    ref("save", P("in.out"), P("in.out.save.out"), "out")
    ref("load", P("in.bal"), P("in.bal.load.txt"))
    ref("save", P("in.bal"), P("in.bal.save.bal"), "bal")
    ref("rip", P("in_rip.nvm"), P("in_rip.rip.txt"))
    size = sum(os.path.getsize(P(f)) for f in os.listdir(OUT))
    print("%d files, %d KiB under %s" % (len(os.listdir(OUT)), size // 1024, OUT))


if __name__ == "__main__":
    main()
