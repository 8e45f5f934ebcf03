"""Shadows the reference's activation.py."""
from nerf_signature_amd.activation import trunc_exp  # noqa: F401
