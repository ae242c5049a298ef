"""`from tetra_sampler.lbs import batch_rodrigues` (lib/smplman.py:16)."""
from d3ga_amd.cage_deform import batch_rodrigues  # noqa: F401

__all__ = ["batch_rodrigues"]
