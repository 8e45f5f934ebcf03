"""Shadows the reference's `raymarching` package (raymarching/__init__.py:1, raymarching/raymarching.py)."""
from nerf_signature_amd.raymarching import *  # noqa: F401,F403
from nerf_signature_amd.raymarching import (composite_rays, composite_rays_train, march_rays, march_rays_train, morton3D,  # noqa: F401
                                            morton3D_invert, near_far_from_aabb, packbits, sph_from_ray)
