import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def hostcheck():
    """CPU build of the product's per-element arithmetic headers (tests/hostcheck/hostcheck.cpp)."""
    src = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")
    out_dir = os.path.join(ROOT, "tests", "hostcheck", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libhostcheck.so")
    deps = [src] + [os.path.join(ROOT, "d3ga_amd", "csrc", h) for h in ("d3ga_math.h", "raster_pre_body.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    return ctypes.CDLL(so)


def ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)
