"""Shadows the reference's nerf/renderer.py (the clean renderer is renderer_wtmk.py minus the message argument)."""
from nerf_signature_amd.renderer import NeRFRenderer, custom_meshgrid  # noqa: F401
