import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

# puts raymarching / gridencoder / freqencoder / shencoder (and their _backend twins) on sys.path
PKG = importlib.import_module("stable-dreamfusion_amd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def hostmath():
    """tests/hostmath: the device arithmetic header built for the host (g++, no contraction)."""
    import ctypes
    d = os.path.join(TESTS, "hostmath")
    so = os.path.join(d, "libhostmath.so")
    src = os.path.join(d, "hostmath.cpp")
    hdrs = [os.path.join(ROOT, "stable-dreamfusion_amd", "csrc", h) for h in ("sdfx_math.h", "shade_math.h", "optim_math.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")
