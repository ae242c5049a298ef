"""Drop-in `tetra_sampler` package surface used by D3GA (lib/cage.py:17), served by d3ga_amd.

Only the pieces on or next to the deform hot path are provided (Tetra, compute_bary).  `body_model.SMPLlayer`
and `lbs.batch_rodrigues` (lib/smplman.py:9,16) need the licensed SMPL-X assets and are out of scope."""
from d3ga_amd.tetra import Tetra, compute_bary  # noqa: F401

__all__ = ["Tetra", "compute_bary"]
