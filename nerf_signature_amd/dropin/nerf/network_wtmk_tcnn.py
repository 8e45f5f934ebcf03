"""Shadows the reference's nerf/network_wtmk_tcnn.py (`from nerf.network_wtmk_tcnn import NeRFNetwork`, main_nerf_wtmk.py:87)."""
from nerf_signature_amd.network import NeRFNetwork as _NeRFNetwork


class NeRFNetwork(_NeRFNetwork):
    """The same model; under the reference's own Trainer (eager loop, GradScaler, torch.optim.Adam) the selected codebook tables' D identical gradients
    are kept as ONE shared tensor and their Adam step runs as one fused pass (the attribute shared_gradient_step; False: plain autograd gradients and the
    optimiser's own loop), and the watermark-block rays -- the same two tensors every step -- take the kept-planes route from their second sighting on
    (the attribute auto_fix_rays)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.shared_gradient_step = True
        self.auto_fix_rays = True
