"""HiDDeN-style watermark decoder (mirror of /root/reference/nerf/hidden_models.py:13-35,104-137,181-183).

This part of the path stays on stock PyTorch-ROCm operators (MIOpen convolutions): SURVEY.md 8(a) R13.
Module/parameter names reproduce the reference's state_dict keys
(`layers.{0..8}.layers.{0,1}.{weight,bias}`, `linear.{weight,bias}`)."""
import torch
import torch.nn as nn

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
