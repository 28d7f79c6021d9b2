import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    name = "large-scale-sfm_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


def scene_from_golden(name):
    """Scene (inputs) + dict of reference outputs from tests/golden/ref_<name>.npz."""
    z = np.load(os.path.join(GOLDEN, f"ref_{name}.npz"))
    scene = pkg("scene")
    sc = scene.Scene(n=int(z["n"]), m=int(z["m"]), cnp=int(z["cnp"]), est_focal=int(z["est_focal"]),
                     undistort=int(z["undistort"]), explicit_centers=int(z["explicit_centers"]),
                     proj_sign=float(z["proj_sign"]), pt_ptr=z["pt_ptr"], obs_cam=z["obs_cam"], x=z["x"],
                     R0=z["R0"], f_fixed=z["f_fixed"], p0=z["p0"], p_true=z["p0"], name=name)
    return sc, z


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()
