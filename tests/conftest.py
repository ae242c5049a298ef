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


@pytest.fixture(autouse=True, scope="session")
def _poison_uninitialised_memory():
    """D3GA_POISON=1: every torch.empty / empty_like on the GPU comes back filled with 0xFF bytes (NaN as float,
    -1 / 4e9 as integers), so that any kernel that reads a buffer before writing it -- and only works because fresh
    allocations happen to hold zeros or last step's values -- fails loudly.  Run: D3GA_POISON=1 pytest -m gpu."""
    if os.environ.get("D3GA_POISON") != "1":
        yield
        return
    import torch
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poison(t):
        if t.is_cuda and t.numel() > 0 and t.is_contiguous():
            if t.dtype == torch.bool:
                t.fill_(True)
            else:
                t.reshape(-1).view(torch.uint8).fill_(0xFF)
        return t

    torch.empty = lambda *a, **k: poison(real_empty(*a, **k))
    torch.empty_like = lambda *a, **k: poison(real_empty_like(*a, **k))
    yield
    torch.empty, torch.empty_like = real_empty, real_empty_like
