"""nerf_signature_amd: the MI355X-native watermarked-NeRF render path (hand-written HIP kernels behind a C ABI,
driven from PyTorch-ROCm) with the reference's Python surface.  See DESIGN.md."""
from . import _native  # noqa: F401  (loads lazily; raises if libnerfsig.so is missing when first used)

__all__ = ["raymarching", "fieldops", "network", "renderer", "trainer", "synthetic", "dp"]
