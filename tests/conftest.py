import importlib
import os
import sys

import pytest

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
