"""Shadows the reference's nerf/renderer_wtmk.py."""
from nerf_signature_amd.renderer import NeRFRenderer, custom_meshgrid  # noqa: F401
