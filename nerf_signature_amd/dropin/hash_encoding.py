"""Shadows the reference's hash_encoding.py."""
from nerf_signature_amd.hash_encoding import HashEmbedder, SHEncoder  # noqa: F401
