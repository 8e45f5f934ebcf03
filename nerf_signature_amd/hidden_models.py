"""HiDDeN-style watermark decoder (mirror of /root/reference/nerf/hidden_models.py:13-35,104-137,181-183).

The convolutions stay on MIOpen; on the GPU the BatchNorm(batch statistics)+GELU pair of every
block is one libnerfsig kernel each way (dec_bn_gelu_fwd/_bwd) instead of torch's 5 + 6 launch-bound kernels, and the
convolution's bias -- which BatchNorm's mean subtraction cancels exactly -- is not added (its gradient is identically zero).
Module/parameter names reproduce the reference's state_dict keys
(`layers.{0..8}.layers.{0,1}.{weight,bias}`, `linear.{weight,bias}`)."""
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

    def forward(self, img_w):
        x = self.layers(img_w).squeeze(-1).squeeze(-1)
        x = self.linear(x)
        x = x.view(-1, self.num_bits, self.redundancy)
        return torch.sum(x, dim=-1)


def get_hidden_decoder_multi_views(num_bits, redundancy=1, num_blocks=7, input_ch=3, channels=64):
    return HiddenDecoder_multi_views(num_blocks=num_blocks, num_bits=num_bits, input_ch=input_ch, channels=channels, redundancy=redundancy)
