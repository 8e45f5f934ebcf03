"""Shadows the reference's nerf/network_wtmk_tcnn.py (`from nerf.network_wtmk_tcnn import NeRFNetwork`, main_nerf_wtmk.py:87)."""
from nerf_signature_amd.network import NeRFNetwork  # noqa: F401
