"""The distortion layer of the training step: Trainer.distortion_layer of /root/reference/nerf/utils_wtmk_disen.py:551-577, applied to the
clamped block renders in front of the decoder (:594), selected by `--distortion` (main_nerf_wtmk.py:75).

    noise       x + n,  n ~ N(0, 0.1) per element (`torch.normal(0, sqrt(0.1), size)`)
    brightness  torchvision ColorJitter(brightness=0.5): ONE factor f ~ U[0.5, 1.5] per call, clamp(f * x, 0, 1)
    blurring    torchvision GaussianBlur(kernel_size=3, sigma=(0.01, 0.5)): ONE sigma ~ U[0.01, 0.5] per call, separable taps
                exp(-0.5 (d / sigma)^2), d in {-1, 0, 1}, normalised; reflect padding
    rotation    torchvision RandomRotation((-30, 30)) per image: nearest-neighbour resampling about the centre, zero fill
    scaling     per image [3, H, W]: F.interpolate(scale_factor=sf ~ U[0.75, 1.25], mode='linear') -- 1-d, along W only

noise / brightness / blurring run inside the fused decoder's first layer (dec_forward_train: no launch of their own) or, for decoder
shapes the fused chain does not implement, as wm_distort_fwd / _bwd; rotation / scaling change the sampling geometry and are one small
launch in front of the decoder and one behind its backward (wm_distort_geom_fwd / _bwd).  The random draws live in device buffers so that a
captured step (trainer.GraphedWatermarkLoop) refreshes them itself from a counter-based generator (wm_distort_draw, keyed by the replay
count); the eager loop draws like the reference's libraries do, from torch's generators.  The scaling factor decides the decoder's input
WIDTH (floor(W * sf)), i.e. a tensor shape: it is always drawn by the host.
torchvision is not installed here: rotation restates its documented resampling and is pinned by nothing."""
import math

import torch
import torch.nn.functional as F

from . import _native as nv

KINDS = {"none": 0, "noise": 1, "brightness": 2, "blurring": 3, "rotation": 4, "scaling": 5}


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


class _DistortGeometry(torch.autograd.Function):
    """rotation / scaling of clamp(img) for [B, H, W, C] (img = the unclamped render) -> (distorted [B, H, W_out, C], clamp(img))."""

    @staticmethod
    def forward(ctx, img, kind, param, out_width):
        img = img.contiguous().float()
        B, H, W, C = img.shape
        out = torch.empty(B, H, out_width, C, dtype=torch.float32, device=img.device)
        clamped = torch.empty_like(img)
        nv.call("wm_distort_geom_fwd", nv.ptr(img), B, H, W, C, kind, nv.ptr(param), out_width, nv.ptr(out), nv.ptr(clamped), nv.stream())
        ctx.save_for_backward(img, param)
        ctx.kind, ctx.out_width = kind, out_width
        ctx.mark_non_differentiable(clamped)
        ctx.set_materialize_grads(False)
        return out, clamped

    @staticmethod
    def backward(ctx, g, _=None):
        if g is None:
            return None, None, None, None
        img, param = ctx.saved_tensors
        B, H, W, C = img.shape
        gx = torch.empty_like(img)
        nv.call("wm_distort_geom_bwd", nv.ptr(g.contiguous().float()), nv.ptr(img), B, H, W, C, ctx.kind, nv.ptr(param), ctx.out_width, nv.ptr(gx), nv.stream())
        return gx, None, None, None


def host_uniform(seed, step, lo, hi):
    """One U[lo, hi) draw as a pure function of (seed, step) -- splitmix64's finaliser, the generator the device-side draws use -- for the one parameter a
    captured loop needs on the HOST (the scaling factor: it decides which captured shape a step replays); the same on every rank."""
    mask = (1 << 64) - 1

    def mix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        return z ^ (z >> 31)

    key = mix((int(seed) ^ (0x9E3779B97F4A7C15 * (int(step) + 1))) & mask)
    u = (mix(key ^ mask) >> 40) / 16777216.0
    return lo + (hi - lo) * u


def scaled_width(W, sf):
    """F.interpolate's output size for scale_factor = sf: floor(W * sf) in double precision, at least one column."""
    return max(1, int(math.floor(float(W) * float(sf))))


class DistortionLayer:
    """distortion: one of none | noise | brightness | blurring | rotation | scaling.

    draw(shape, device)              the eager loop's per-step draws: torch's generators, the calls the reference's libraries make (from generators of
                                     the layer's own, seeded with `seed`)
    draw_on_device(step_counter)     the captured loop's: wm_distort_draw from (seed, replay count) -- no host value involved
    param / noise                    the device buffers the decoder reads (static addresses: a captured graph holds them)
    out_width(W)                     the decoder's input width for this step's draw (scaling: floor(W * sf); W otherwise)
    __call__(pred_rgb)               the layer on the CLAMPED blocks [B, H, W, 3] as stand-alone operators (rotation, scaling, and the
                                     other kinds when the decoder's fused first layer is not available)"""

    def __init__(self, distortion="none", seed=0):
        if distortion not in KINDS:
            raise ValueError(f"distortion {distortion!r}: choose from none, noise, rotation, scaling, blurring, brightness (main_nerf_wtmk.py:75)")
        self.name, self.kind, self.seed = distortion, KINDS[distortion], int(seed)
        self.param = self.noise = None
        self.factor = 1.0             # scaling: this step's factor as the host knows it (it decides a shape)
        # generators of the layer's own, seeded alike on every rank: the decoder is replicated in a data-parallel run (it sees the all-gathered
        # blocks), so every rank has to distort them the same way
        self._host_gen = torch.Generator(device="cpu").manual_seed(self.seed)
        self._dev_gen = None

    @property
    def native(self):
        return self.kind > 0

    @property
    def fused(self):
        """applied inside the fused decoder's first layer (dec_forward_train)"""
        return 1 <= self.kind <= 3

    @property
    def geometric(self):
        """a resampling kernel of its own in front of the decoder (wm_distort_geom_fwd)"""
        return self.kind >= 4

    def out_width(self, W):
        return scaled_width(W, self.factor) if self.kind == 5 else int(W)

    def _buffers(self, shape, device):
        n = 2 * int(shape[0]) if self.kind == 4 else 1       # rotation: (cos, sin) per image
        if self.param is None or self.param.device != device or self.param.numel() != n:
            self.param = torch.zeros(n, dtype=torch.float32, device=device)
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
        elif self.kind == 3:      # GaussianBlur.get_params: torch.empty(1).uniform_(sigma_min, sigma_max).item()
            self.param.fill_(float(torch.empty(1).uniform_(0.01, 0.5, generator=self._host_gen)))
        elif self.kind == 4:      # RandomRotation.get_params, once per image (utils_wtmk_disen.py:560): float(torch.empty(1).uniform_(-30, 30))
            angles = [math.radians(float(torch.empty(1).uniform_(-30.0, 30.0, generator=self._host_gen))) for _ in range(int(shape[0]))]
            self.set_rotation(angles)
        else:                     # utils_wtmk_disen.py:563: sf = torch.empty(1).uniform_(0.75, 1.25).item(), ONE factor per call
            self.set_scaling(torch.empty(1).uniform_(0.75, 1.25, generator=self._host_gen).item())

    def set_rotation(self, radians):
        """this step's angles, one per image (host values -> the device buffer the kernels read)"""
        self.param.copy_(torch.tensor([[math.cos(a), math.sin(a)] for a in radians], dtype=torch.float32).reshape(-1))

    def set_scaling(self, sf):
        self.factor = float(sf)
        self.param.fill_(self.factor)

    def draw_on_device(self, step_counter, shape, device):
        """The captured loop's draws (noise, brightness, blurring, rotation): a pure function of (seed, *step_counter) computed on the device.  The scaling
        factor is a host value (it selects the captured shape): trainer.GraphedWatermarkLoop sets it through host_uniform / set_scaling."""
        if self.kind == 5:
            raise NotImplementedError("the scaling factor decides the decoder's input width: it is drawn by the host (host_uniform, set_scaling)")
        self._buffers(shape, device)
        n = self.noise.numel() if self.kind == 1 else (int(shape[0]) if self.kind == 4 else 0)
        nv.call("wm_distort_draw", self.kind, self.seed, nv.ptr(step_counter), n, nv.ptr(self.param), nv.ptr(self.noise), nv.stream())

    def __call__(self, pred_rgb, raw=None):
        """pred_rgb: clamped blocks [B, H, W, 3].  raw: the unclamped render (native kinds back-propagate through the clamp themselves)."""
        if self.name == "none":
            return pred_rgb
        if not pred_rgb.is_cuda:
            return reference_ops(pred_rgb, self.name, self.param if self.kind != 5 else self.factor, self.noise)
        if self.geometric:
            return _DistortGeometry.apply(pred_rgb if raw is None else raw, self.kind, self.param, self.out_width(pred_rgb.shape[2]))[0]
        return _Distort.apply(pred_rgb if raw is None else raw, self.kind, self.param, self.noise)


def gaussian_kernel(sigma, dtype=torch.float32, device="cpu"):
    d = torch.tensor([-1.0, 0.0, 1.0], dtype=dtype, device=device)
    k = torch.exp(-0.5 * (d / sigma) ** 2)
    k = k / k.sum()
    return k[:, None] * k[None, :]


def reference_ops(pred_rgb, name, param=None, noise=None):
    """The layer on the CLAMPED blocks [B, H, W, C] as stock operators with the draws passed in (CPU / differentiable; what the kernels are tested
    against).  param: brightness factor / blur sigma (one element), rotation: (cos, sin) per image [2 B], scaling: the factor (a Python float)."""
    if name == "noise":
        return pred_rgb + noise
    if name == "brightness":
        return torch.clamp(pred_rgb * param.reshape(()), 0, 1)
    x = pred_rgb.permute(0, 3, 1, 2)
    if name == "rotation":
        cs = param.detach().reshape(-1, 2).cpu()
        return torch.stack([rotate_nearest(img, float(cs[i, 0]), float(cs[i, 1])) for i, img in enumerate(x)]).permute(0, 2, 3, 1)
    if name == "scaling":         # utils_wtmk_disen.py:565: every image [3, H, W] is a batch of 3 one-dimensional signals with H channels: resized along W only
        return torch.stack([F.interpolate(img, scale_factor=float(param), mode="linear") for img in x]).permute(0, 2, 3, 1)
    C = x.shape[1]
    k = gaussian_kernel(float(param), x.dtype, x.device).expand(C, 1, 3, 3)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), k, groups=C)
    return y.permute(0, 2, 3, 1)


def rotate_nearest(img, cos, sin):
    """[C, H, W] rotated counter-clockwise about its centre by the angle with the given (float32-representable) cosine and sine, same size,
    nearest-neighbour sampling (round half to even), zeros outside: the source of every output pixel centre (x, y measured from the image
    centre) is  x' = cos * x - sin * y,  y' = sin * x + cos * y  -- float32, every product and sum rounded on its own, as the kernel computes it."""
    C, H, W = img.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=img.device) - (H - 1) / 2,
                            torch.arange(W, dtype=torch.float32, device=img.device) - (W - 1) / 2, indexing="ij")
    sx = cos * xs - sin * ys + (W - 1) / 2
    sy = sin * xs + cos * ys + (H - 1) / 2
    ix, iy = torch.round(sx).long(), torch.round(sy).long()
    inside = (ix >= 0) & (ix < W) & (iy >= 0) & (iy < H)
    flat = img.reshape(C, H * W)
    out = flat[:, (iy.clamp(0, H - 1) * W + ix.clamp(0, W - 1)).reshape(-1)].reshape(C, H, W)
    return out * inside.to(img.dtype)
