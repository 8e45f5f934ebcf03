"""The distortion layer of the training step: Trainer.distortion_layer of /root/reference/nerf/utils_wtmk_disen.py:551-577, applied to the
clamped block renders in front of the decoder (:594), selected by `--distortion` (main_nerf_wtmk.py:75).

    noise       x + n,  n ~ N(0, 0.1) per element (`torch.normal(0, sqrt(0.1), size)`)
    brightness  torchvision ColorJitter(brightness=0.5): ONE factor f ~ U[0.5, 1.5] per call, clamp(f * x, 0, 1)
    blurring    torchvision GaussianBlur(kernel_size=3, sigma=(0.01, 0.5)): ONE sigma ~ U[0.01, 0.5] per call, separable taps
                exp(-0.5 (d / sigma)^2), d in {-1, 0, 1}, normalised; reflect padding
    rotation    torchvision RandomRotation((-30, 30)) per image: nearest-neighbour resampling about the centre, zero fill
    scaling     per image [3, H, W]: F.interpolate(scale_factor=sf ~ U[0.75, 1.25], mode='linear') -- 1-d, along W only

noise / brightness / blurring run inside the fused decoder's first layer (dec_forward_distorted: no launch of their own) or, for decoder
shapes the fused chain does not implement, as wm_distort_fwd / _bwd; their random draws live in device buffers so that a captured step
(trainer.GraphedWatermarkLoop) refreshes them itself from a counter-based generator (wm_distort_draw, keyed by the replay count).  The
eager loop draws like the reference's libraries do, from torch's generators.  rotation / scaling change the sampling geometry (scaling
even the decoder's input width, so nothing of it can be captured with static shapes): stock operators, eager loop only.
torchvision is not installed here: rotation restates its documented resampling and is pinned by nothing."""
import math

import torch
import torch.nn.functional as F

from . import _native as nv

KINDS = {"none": 0, "noise": 1, "brightness": 2, "blurring": 3}
STOCK = ("rotation", "scaling")


class _Distort(torch.autograd.Function):
    """D(clamp(img)) for the native kinds on [B, H, W, C] (img = the unclamped render)."""

    @staticmethod
    def forward(ctx, img, kind, param, noise):
        img = img.contiguous().float()
        B, H, W, C = img.shape
        out = torch.empty_like(img)
        nv.call("wm_distort_fwd", nv.ptr(img), B, H, W, C, kind, nv.ptr(param), nv.ptr(noise), nv.ptr(out), nv.stream())
        ctx.save_for_backward(img, param, noise)
        ctx.kind = kind
        return out

    @staticmethod
    def backward(ctx, g):
        img, param, noise = ctx.saved_tensors
        B, H, W, C = img.shape
        gx = torch.empty_like(img)
        nv.call("wm_distort_bwd", nv.ptr(g.contiguous().float()), nv.ptr(img), B, H, W, C, ctx.kind, nv.ptr(param), nv.ptr(noise), nv.ptr(gx), nv.stream())
        return gx, None, None, None


class DistortionLayer:
    """distortion: one of none | noise | brightness | blurring | rotation | scaling.

    draw(shape, device)              the eager loop's per-step draws: torch's generators, the calls the reference's libraries make (from generators of
                                     the layer's own, seeded with `seed`)
    draw_on_device(step_counter)     the captured loop's: wm_distort_draw from (seed, replay count) -- no host value involved
    param / noise                    the device buffers the decoder reads (static addresses: a captured graph holds them)
    __call__(pred_rgb)               the layer on the CLAMPED blocks [B, H, W, 3] as stock / stand-alone operators (rotation, scaling,
                                     and the native kinds when the decoder's fused first layer is not available)"""

    def __init__(self, distortion="none", seed=0):
        if distortion not in KINDS and distortion not in STOCK:
            raise ValueError(f"distortion {distortion!r}: choose from none, noise, rotation, scaling, blurring, brightness (main_nerf_wtmk.py:75)")
        self.name, self.kind, self.seed = distortion, KINDS.get(distortion, -1), int(seed)
        self.param = self.noise = None
        # generators of the layer's own, seeded alike on every rank: the decoder is replicated in a data-parallel run (it sees the all-gathered
        # blocks), so every rank has to distort them the same way
        self._host_gen = torch.Generator(device="cpu").manual_seed(self.seed)
        self._dev_gen = None

    @property
    def native(self):
        return self.kind > 0

    def _buffers(self, shape, device):
        if self.param is None or self.param.device != device:
            self.param = torch.zeros(1, dtype=torch.float32, device=device)
        if self.kind == 1 and (self.noise is None or tuple(self.noise.shape) != tuple(shape) or self.noise.device != device):
            self.noise = torch.zeros(tuple(shape), dtype=torch.float32, device=device)

    def draw(self, shape, device):
        if not self.native:
            return
        self._buffers(shape, device)
        if self.kind == 1:        # utils_wtmk_disen.py:555: torch.normal(0, sqrt(0.1), size=pred_rgb.shape, device=pred_rgb.device)
            if self._dev_gen is None or self._dev_gen.device != device:
                self._dev_gen = torch.Generator(device=device).manual_seed(self.seed)
            torch.normal(0.0, math.sqrt(0.1), size=tuple(shape), generator=self._dev_gen, device=device, out=self.noise)
        elif self.kind == 2:      # ColorJitter.get_params: float(torch.empty(1).uniform_(lo, hi)) on the host
            self.param.fill_(float(torch.empty(1).uniform_(0.5, 1.5, generator=self._host_gen)))
        else:                     # GaussianBlur.get_params: torch.empty(1).uniform_(sigma_min, sigma_max).item()
            self.param.fill_(float(torch.empty(1).uniform_(0.01, 0.5, generator=self._host_gen)))

    def draw_on_device(self, step_counter, shape, device):
        if not self.native:
            raise NotImplementedError(f"distortion {self.name!r} runs on stock operators with per-step host draws (and 'scaling' changes the decoder's input "
                                      f"shape): drive it with the eager WatermarkLoop")
        self._buffers(shape, device)
        n = self.noise.numel() if self.kind == 1 else 0
        nv.call("wm_distort_draw", self.kind, self.seed, nv.ptr(step_counter), n, nv.ptr(self.param), nv.ptr(self.noise), nv.stream())

    def __call__(self, pred_rgb, raw=None):
        """pred_rgb: clamped blocks [B, H, W, 3].  raw: the unclamped render (native kinds back-propagate through the clamp themselves)."""
        if self.name == "none":
            return pred_rgb
        if self.native:
            if pred_rgb.is_cuda:
                return _Distort.apply(pred_rgb if raw is None else raw, self.kind, self.param, self.noise)
            return reference_ops(pred_rgb, self.name, self.param, self.noise)
        x = pred_rgb.permute(0, 3, 1, 2)
        if self.name == "rotation":
            out = torch.stack([rotate_nearest(img, float(torch.empty(1).uniform_(-30.0, 30.0, generator=self._host_gen))) for img in x])
        else:
            sf = torch.empty(1).uniform_(0.75, 1.25, generator=self._host_gen).item()        # utils_wtmk_disen.py:563: ONE factor per call, every image [3, H, W] resized along W
            out = torch.stack([F.interpolate(img, scale_factor=sf, mode="linear") for img in x])
        return out.permute(0, 2, 3, 1)


def gaussian_kernel(sigma, dtype=torch.float32, device="cpu"):
    d = torch.tensor([-1.0, 0.0, 1.0], dtype=dtype, device=device)
    k = torch.exp(-0.5 * (d / sigma) ** 2)
    k = k / k.sum()
    return k[:, None] * k[None, :]


def reference_ops(pred_rgb, name, param=None, noise=None):
    """The native kinds as stock operators with the draws passed in (CPU / differentiable; what the kernels are tested against)."""
    if name == "noise":
        return pred_rgb + noise
    if name == "brightness":
        return torch.clamp(pred_rgb * param.reshape(()), 0, 1)
    x = pred_rgb.permute(0, 3, 1, 2)
    C = x.shape[1]
    k = gaussian_kernel(float(param), x.dtype, x.device).expand(C, 1, 3, 3)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), k, groups=C)
    return y.permute(0, 2, 3, 1)


def rotate_nearest(img, degrees):
    """[C, H, W] rotated counter-clockwise by `degrees` about its centre, same size, nearest-neighbour sampling, zeros outside: the inverse
    map of every output pixel centre (x, y measured from the image centre) is  x' = cos * x - sin * y,  y' = sin * x + cos * y."""
    C, H, W = img.shape
    a = math.radians(degrees)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=img.device) - (H - 1) / 2,
                            torch.arange(W, dtype=torch.float32, device=img.device) - (W - 1) / 2, indexing="ij")
    sx = math.cos(a) * xs - math.sin(a) * ys + (W - 1) / 2
    sy = math.sin(a) * xs + math.cos(a) * ys + (H - 1) / 2
    ix, iy = torch.round(sx).long(), torch.round(sy).long()
    inside = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
    flat = img.reshape(C, H * W)
    out = flat[:, (iy.clamp(0, H - 1) * W + ix.clamp(0, W - 1)).reshape(-1)].reshape(C, H, W)
    return out * inside.to(img.dtype)
