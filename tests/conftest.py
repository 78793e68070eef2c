import importlib
import os
import sys

import pytest

# PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 next to /opt/rocm's, which libhprt.so links against; one process can
# hold both, but only if torch's are loaded first (a torch.cuda call after libhprt has opened the device reports "No HIP GPUs are
# available").  Tests that hand torch device buffers to the library (as bench.py does) need both, so torch is imported before
# anything loads libhprt.so — whatever subset of the test files runs.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
KILLEROO = os.path.join(GOLDEN, "killeroo_simple.hprt")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hprt():
    """The product package; builds libhprt.so in-tree if it is missing."""
    lib = os.path.join(ROOT, "thesis-pbrt-v3_amd", "lib", "libhprt.so")
    if not os.path.exists(lib):
        sys.path.insert(0, os.path.join(ROOT, "thesis-pbrt-v3_amd"))
        import build as hprt_build
        hprt_build.build()
    return importlib.import_module("thesis-pbrt-v3_amd")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    return _orc


@pytest.fixture(scope="session")
def killeroo_model(hprt):
    return hprt.Model.load(KILLEROO)


@pytest.fixture(scope="session")
def killeroo_bvh(hprt, killeroo_model):
    return hprt.Bvh(killeroo_model)


@pytest.fixture(scope="session")
def killeroo_oracle(orc):
    return orc.OracleScene(KILLEROO)


@pytest.fixture(scope="session")
def killeroo_scene(hprt, killeroo_model, killeroo_bvh):
    """Device scene; only -m gpu tests request it."""
    return hprt.Scene(killeroo_model, killeroo_bvh)
