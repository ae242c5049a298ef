"""`from tetra_sampler.body_model import SMPLlayer` (lib/smplman.py:9): importable, not constructible without SMPL-X."""
from d3ga_amd.cage_deform import SMPLlayer  # noqa: F401

__all__ = ["SMPLlayer"]
