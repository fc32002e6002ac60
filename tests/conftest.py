import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "omnihuman-1-hack_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def omh():
    """The product package (loads libomh.so; builds it in-tree if missing)."""
    return importlib.import_module(PKG)


@pytest.fixture(scope="session")
def ops(omh):
    return importlib.import_module(PKG + ".ops")


@pytest.fixture(scope="session")
def wan_model_mod(omh):
    return importlib.import_module(PKG + ".wan.modules.model")


def rel_rms(a, b):
    """||a-b|| / ||b|| in fp64."""
    import torch
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def set_option(key, value):
    """Flip a dispatch switch of libomh.so for the running test (include/omh.h: omh_set_option; ``value`` None unsets).
    The library reads the environment once per process, so the tests change switches through the ABI, not through
    os.environ; the autouse fixture below restores every switch after each test."""
    importlib.import_module(PKG + ".ops").set_option(key, value)


@pytest.fixture(autouse=True)
def _restore_library_options():
    yield
    mod = sys.modules.get(PKG + ".ops")
    if mod is not None:
        mod.reset_options()
