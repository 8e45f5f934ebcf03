"""Shadows the reference's nerf/network_hash.py (stage-1 clean model, `from nerf.network_hash import NeRFNetwork`)."""
from nerf_signature_amd.stage1 import CleanNeRFNetwork as NeRFNetwork  # noqa: F401
