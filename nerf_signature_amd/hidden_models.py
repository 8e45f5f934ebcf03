"""HiDDeN-style watermark decoder (mirror of /root/reference/nerf/hidden_models.py:13-35,104-137,181-183).

On the GPU the decoder of the watermark path (9 ConvBNRelu blocks, 64 channels, one output bit) runs as libnerfsig's fused
chain (dec_forward/dec_backward: one kernel per layer each way, csrc/decoder_fused.hip) instead of ~120 launch-bound
stock operators per step.  Other shapes fall back to MIOpen convolutions with the BatchNorm(batch statistics)+GELU pair
fused (dec_bn_gelu_fwd/_bwd).  Either way a convolution's bias -- which BatchNorm's mean subtraction cancels exactly -- is
not added, and its gradient (identically zero) is reported as None.  NERFSIG_DECODER=torch forces the stock operator chain.
Module/parameter names reproduce the reference's state_dict keys
(`layers.{0..8}.layers.{0,1}.{weight,bias}`, `linear.{weight,bias}`)."""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as nv

_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


_CONST = {}


def _mean_std(x):
    """Per-(device, dtype) constants, created once: a host-to-device copy inside a captured stream is not allowed."""
    key = (x.device, x.dtype)
    if key not in _CONST:
        _CONST[key] = (torch.tensor(_MEAN, dtype=x.dtype, device=x.device).view(-1, 1, 1),
                       torch.tensor(_STD, dtype=x.dtype, device=x.device).view(-1, 1, 1))
    return _CONST[key]


def normalize_img(x):
    """torchvision.transforms.Normalize(mean, std) of hidden_models.py:13 for [B,3,H,W] or [3,H,W] tensors."""
    mean, std = _mean_std(x)
    return (x - mean) / std


def unnormalize_img(x):
    mean, std = _mean_std(x)
    return x * std + mean


def _nhwc(t):
    """Device pointer of a 4-d tensor held channels-last (nv.ptr insists on row-major contiguity)."""
    import ctypes
    if not (t.is_cuda and t.is_contiguous(memory_format=torch.channels_last) and t.dtype == torch.float32):
        raise ValueError("expected a float32 channels-last device tensor")
    return ctypes.c_void_p(t.data_ptr())


class _BNGelu(torch.autograd.Function):
    """gelu(batch_norm(x; batch statistics, eps)) on an NHWC tensor: one kernel forward, one backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        N, C, H, W = x.shape
        x = x.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x, memory_format=torch.channels_last)
        save = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        nv.call("dec_bn_gelu_fwd", _nhwc(x), nv.ptr(gamma), nv.ptr(beta), N, C, H * W, eps, _nhwc(y), nv.ptr(save), nv.stream())
        ctx.save_for_backward(x, gamma, beta, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, save = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx, dgamma, dbeta = torch.empty_like(x, memory_format=torch.channels_last), torch.empty_like(gamma), torch.empty_like(beta)
        nv.call("dec_bn_gelu_bwd", _nhwc(dy), _nhwc(x), nv.ptr(gamma), nv.ptr(beta), nv.ptr(save), N, C, H * W, _nhwc(dx), nv.ptr(dgamma),
                nv.ptr(dbeta), nv.stream())
        return dx, dgamma, dbeta, None


_WEIGHTS_STREAM = None


def set_weights_stream(stream):
    """The stream the fused decoder's parameter-gradient kernels are queued on (None: the stream of the backward itself).  Whoever
    sets one joins it before the optimiser reads the gradients (trainer.GraphedWatermarkLoop does)."""
    global _WEIGHTS_STREAM
    _WEIGHTS_STREAM = stream


_GRAD_ARENA = None


def set_grad_arena(buffer):
    """A float32 device buffer the fused decoder's backward carves its flat parameter-gradient block from (its first n elements)
    instead of allocating one.  The data-parallel loop hands in the tail of the allocation that also holds the shared codebook
    gradient G, so that one all-reduce covers both (dp.GradExchange).  None: allocate per backward."""
    global _GRAD_ARENA
    _GRAD_ARENA = buffer


class _FusedDecoder(torch.autograd.Function):
    """decoded = Linear(AvgPool(ConvBNRelu^9(img))) through dec_forward / dec_backward.  params: for each of the 9 blocks
    (conv weight, bn weight, bn bias), then the linear weight and bias.
    rendered=False: img is the normalised [B,Cin,H,W] tensor.  rendered=True: img is the compositor's [B,H,W,Cin] output and the
    clamp / permute / normalize_img of the training step happen inside layer 0; the second output is the clamped image."""

    @staticmethod
    def forward(ctx, img, eps, rendered, distortion, bce, *params):
        """distortion: a distortion.DistortionLayer of a fused kind (noise / brightness / blurring) whose draws are in its device buffers (rendered=True only), or None.
        bce: (message [B] float32 on the device, temp, scale) or None (rendered=True only): the head kernel also leaves
        scale * (sigmoid(temp * decoded) - message) in `_FusedDecoder.seed` -- the watermark loss's gradient, ready for this node's backward."""
        img = img.contiguous()
        if rendered:
            B, H, W, Cin = img.shape
            mean, std = (ctypes.c_float * Cin)(*_MEAN[:Cin]), (ctypes.c_float * Cin)(*_STD[:Cin])
            clamped = torch.empty_like(img)
        else:
            B, Cin, H, W = img.shape
            mean = std = clamped = None
        ps = [p.detach().contiguous() for p in params]
        ws = torch.empty(nv.fn("dec_workspace_bytes")(B, Cin, H, W), dtype=torch.uint8, device=img.device)
        out = torch.empty(B, dtype=torch.float32, device=img.device)
        dist = distortion if (distortion is not None and distortion.fused) else None
        _FusedDecoder.seed = None
        if dist is not None or bce is not None:
            if not rendered:
                raise ValueError("the fused distortion layer / loss seed act on the rendered blocks (rendered=True)")
            seed = None
            if bce is not None:
                msg, temp, scale = bce
                if not (msg.is_cuda and msg.dtype == torch.float32 and msg.numel() == B and msg.is_contiguous()):
                    raise ValueError("bce: message must be a contiguous float32 device tensor with one entry per image")
                seed = torch.empty(B, dtype=torch.float32, device=img.device)
            nv.call("dec_forward_train", nv.ptr(img), mean, std, nv.ptr_array(ps), B, Cin, H, W, eps, nv.ptr(ws), nv.ptr(out), nv.ptr(clamped),
                    dist.kind if dist is not None else 0, nv.ptr(dist.param) if dist is not None else None, nv.ptr(dist.noise) if dist is not None else None,
                    nv.ptr(msg) if bce is not None else None, float(temp) if bce is not None else 0.0, float(scale) if bce is not None else 0.0, nv.ptr(seed),
                    nv.stream())
            _FusedDecoder.seed = None if seed is None else seed.view(B, 1)      # (the caller takes it right behind this call: trainer.train_step)
        else:
            nv.call("dec_forward", nv.ptr(img), int(rendered), mean, std, nv.ptr_array(ps), B, Cin, H, W, eps, nv.ptr(ws), nv.ptr(out), nv.ptr(clamped),
                    nv.stream())
        ctx.dist = dist
        out = out.view(B, 1)
        ctx.save_for_backward(img, ws, *ps)
        ctx.geom = (B, Cin, H, W, bool(rendered))
        ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the non-differentiable `clamped` output (one fill launch on the backward's critical path)
        if rendered:
            ctx.mark_non_differentiable(clamped)
            return out, clamped
        return out

    @staticmethod
    def backward(ctx, grad_out, *_):
        if grad_out is None:
            return (None,) * (5 + len(ctx.saved_tensors) - 2)
        img, ws, *ps = ctx.saved_tensors
        B, Cin, H, W, rendered = ctx.geom
        mean, std = ((ctypes.c_float * Cin)(*_MEAN[:Cin]), (ctypes.c_float * Cin)(*_STD[:Cin])) if rendered else (None, None)
        # one flat buffer, the parameter gradients are views of it: the data-parallel exchange all-reduces it in place
        # (dp.GradExchange) instead of packing 27 tensors into a bucket and unpacking them again
        n_flat = sum(p.numel() for p in ps)
        arena = _GRAD_ARENA
        if arena is not None and arena.device == img.device and arena.numel() >= n_flat:
            flat = arena[:n_flat]
        else:
            flat = torch.empty(n_flat, dtype=torch.float32, device=img.device)
        grads, off = [], 0
        for p in ps:
            grads.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        grad_img = torch.empty_like(img)
        side = _WEIGHTS_STREAM
        if side is not None:   # the buffers below are allocated on this stream but also read / written on the side stream
            for t in (flat, ws, img, grad_out, *ps):
                t.record_stream(side)
        if ctx.dist is not None:     # (the draws in its buffers are still this step's: they are refreshed at the head of the next one)
            d = ctx.dist
            scratch = torch.empty_like(img) if d.kind == 3 else None
            nv.call("dec_backward_train", nv.ptr(grad_out.contiguous().view(-1)), nv.ptr(img), mean, std, nv.ptr_array(ps), B, Cin, H, W, nv.ptr(ws),
                    nv.ptr_array(grads), nv.ptr(grad_img), d.kind, nv.ptr(d.param), nv.ptr(d.noise), nv.ptr(scratch), nv.stream(),
                    nv.stream() if side is None else side.cuda_stream)
        else:
            nv.call("dec_backward", nv.ptr(grad_out.contiguous().view(-1)), nv.ptr(img), int(rendered), mean, std, nv.ptr_array(ps), B, Cin, H, W,
                    nv.ptr(ws), nv.ptr_array(grads), nv.ptr(grad_img), nv.stream(), nv.stream() if side is None else side.cuda_stream)
        return (grad_img, None, None, None, None, *grads)


class ConvBNRelu(nn.Module):
    """3x3 convolution, BatchNorm that always uses batch statistics (track_running_stats=False), GELU."""

    def __init__(self, channels_in, channels_out):
        super().__init__()
        self.layers = nn.Sequential(
            nn.Conv2d(channels_in, channels_out, 3, stride=1, padding=1),
            nn.BatchNorm2d(channels_out, eps=1e-3, track_running_stats=False),
            nn.GELU(),
        )

    def forward(self, x):
        conv, bn = self.layers[0], self.layers[1]
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] * x.shape[2] * x.shape[3] > 1:
            # BatchNorm subtracts the per-channel batch mean, so conv.bias cancels: skip the add and its (zero) gradient.
            return _BNGelu.apply(F.conv2d(x, conv.weight, None, 1, 1), bn.weight.float(), bn.bias.float(), bn.eps)
        return self.layers(x)


class HiddenDecoder_multi_views(nn.Module):
    def __init__(self, num_blocks, num_bits, input_ch, channels, redundancy=1):
        super().__init__()
        layers = [ConvBNRelu(input_ch, channels)]
        for _ in range(num_blocks - 1):
            layers.append(ConvBNRelu(channels, channels))
        layers.append(ConvBNRelu(channels, num_bits * redundancy))
        layers.append(nn.AdaptiveAvgPool2d(output_size=(1, 1)))
        self.layers = nn.Sequential(*layers)
        self.linear = nn.Linear(num_bits * redundancy, num_bits * redundancy)
        self.num_bits = num_bits
        self.redundancy = redundancy

    def _fused_params(self, B, Cin, H, W, like):
        """The 29 parameters dec_forward takes, or None when this decoder / input is not the shape the fused chain implements."""
        if not (like.is_cuda and like.dtype == torch.float32) or os.environ.get("NERFSIG_DECODER", "") == "torch":
            return None
        key = (B, Cin, H, W, like.device)
        # The structural checks are remembered per (shape, device); the 29 parameters are re-read every call straight from the sub-modules' parameter
        # dicts (a replaced Parameter is then picked up; nn.Module attribute lookups would cost ~0.1 ms per call in an eager loop)
        cached = self.__dict__.get("_fused_cache")
        if cached is None or cached[0] != key:
            blocks = list(self.layers)[:-1]
            if len(blocks) != 9 or self.num_bits * self.redundancy != 1:
                return None
            convs, bns = [b.layers[0] for b in blocks], [b.layers[1] for b in blocks]
            want = [(Cin, 64)] + [(64, 64)] * 7 + [(64, 1)]
            if [(c.in_channels, c.out_channels) for c in convs] != want or len({bn.eps for bn in bns}) != 1:
                return None
            if not nv.fn("dec_workspace_bytes")(B, Cin, H, W):
                return None
            cached = self.__dict__["_fused_cache"] = (key, bns[0].eps, [(c._parameters, bn._parameters) for c, bn in zip(convs, bns)], self.linear._parameters)
        params = []
        for cp, bp in cached[2]:
            params += [cp["weight"], bp["weight"], bp["bias"]]
        params += [cached[3]["weight"], cached[3]["bias"]]
        for p in params:
            if p.dtype != torch.float32 or p.device != like.device:
                return None
        return cached[1], params

    def decode_rendered(self, image, distortion=None, bce=None):
        """bce = (message, temp, scale): see _FusedDecoder.forward (ignored where the fused chain does not run).
        The training step's `msg_decoder(normalize_img(distortion_layer(clamp(image, 0, 1)).permute(0, 3, 1, 2)))` for the compositor's
        [B, H, W, 3] blocks (utils_wtmk_disen.py:592-595) -> (decoded [B, 1], clamped image).  On the GPU the clamp, the distortion
        (noise / brightness / blurring: distortion.DistortionLayer with this step's draws in its buffers), the layout change and the
        normalisation are part of the fused decoder's first layer."""
        geometric = distortion is not None and distortion.geometric
        width = distortion.out_width(image.shape[2]) if geometric and image.dim() == 4 else (image.shape[2] if image.dim() == 4 else 0)
        fused = self._fused_params(image.shape[0], image.shape[3], image.shape[1], width, image) if image.dim() == 4 and image.shape[3] <= 3 else None
        if fused is not None and geometric:
            # rotation / scaling: one resampling launch in front of the chain; its output lies in [0, 1], so the first layer's clamp is the identity
            from .distortion import _DistortGeometry
            resampled, pred = _DistortGeometry.apply(image, distortion.kind, distortion.param, width)
            return _FusedDecoder.apply(resampled, fused[0], True, None, bce, *fused[1])[0], pred
        if fused is not None:
            return _FusedDecoder.apply(image, fused[0], True, distortion if (distortion is not None and distortion.fused) else None, bce, *fused[1])
        pred = torch.clamp(image, min=0, max=1)
        dist = pred if distortion is None else distortion(pred, raw=image)
        return self(normalize_img(dist.permute(0, 3, 1, 2))), pred

    def forward(self, img_w):
        fused = self._fused_params(*img_w.shape, img_w) if img_w.dim() == 4 else None
        if fused is not None:
            return _FusedDecoder.apply(img_w, fused[0], False, None, None, *fused[1])   # num_bits = redundancy = 1: the view/sum below is the identity
        x = self.layers(img_w).squeeze(-1).squeeze(-1)
        x = self.linear(x)
        x = x.view(-1, self.num_bits, self.redundancy)
        return torch.sum(x, dim=-1)


def get_hidden_decoder_multi_views(num_bits, redundancy=1, num_blocks=7, input_ch=3, channels=64):
    return HiddenDecoder_multi_views(num_blocks=num_blocks, num_bits=num_bits, input_ch=input_ch, channels=channels, redundancy=redundancy)
