import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return {n: np.load(os.path.join(GOLDEN, "ref_%s.npz" % n)) for n in ("condense", "lpv", "jac", "ablist")}


@pytest.fixture(scope="session")
def omodel():
    from oracle.model import Model
    return Model.load()


@pytest.fixture(scope="session")
def enc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nmpc_amd
    return nmpc_amd.load_system(device="cuda:0")


def disc_setup(omodel, Nc=10):
    """Normalised bounds / weights of the canonical configuration (NMPC_test.py:326-344)."""
    import numpy as np
    ymin = (np.array([-1.2]) - omodel.y0) / omodel.ystd
    ymax = (np.array([1.2]) - omodel.y0) / omodel.ystd
    umin = (-4 - omodel.u0) / omodel.ustd
    umax = (4 - omodel.u0) / omodel.ustd
    return dict(ymin=ymin, ymax=ymax, umin=umin, umax=umax, Q=np.diag([1000.0, 10.0]), R=1.0, rho=1000.0, Nc=Nc)
