"""d3ga_amd -- MI355X-native deform-and-rasterize hot path of D3GA.

Python host layer over libd3ga_hip.so (hand-written gfx950 HIP kernels behind the C ABI of include/d3ga.h).
There is NO CPU fallback: every op raises if the library is missing or the tensors are not on the GPU.
"""
from ._lib import D3GAError, lib, library_path  # noqa: F401

__all__ = ["D3GAError", "lib", "library_path"]
