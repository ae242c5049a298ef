"""Drop-in `tetra_sampler` package surface used by D3GA, served by d3ga_amd.

`Tetra`, `compute_bary` (lib/cage.py:17) are the pieces on / next to the deform hot path.  The submodules the reference
also imports exist so that its modules load unchanged: `tetra_sampler.lbs.batch_rodrigues` (lib/smplman.py:16, asset-free,
implemented) and `tetra_sampler.body_model.SMPLlayer` (lib/smplman.py:9: importable, raises on construction -- it needs
the licensed SMPL-X assets, out of scope)."""
from d3ga_amd.tetra import Tetra, compute_bary  # noqa: F401

__all__ = ["Tetra", "compute_bary"]
