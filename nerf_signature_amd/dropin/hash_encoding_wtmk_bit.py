"""Shadows the reference's hash_encoding_wtmk_bit.py."""
from nerf_signature_amd.hash_encoding import SHEncoder  # noqa: F401
from nerf_signature_amd.hash_encoding_wtmk_bit import HashEmbedder  # noqa: F401
