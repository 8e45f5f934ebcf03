"""Parameter containers with the surface of the tinycudann modules the reference builds
(/root/reference/nerf/network_wtmk_tcnn.py:52-88): `Network(...).params` is one flat fp32 vector whose
layout (consecutive [out,in] row-major matrices, widths padded to 16) is the one documented in
INTEGRATION.md, so `sigma_net.params` / `color_net.params` / `encoder_dir.params` load key-for-key.
Evaluation does not happen here -- NeRFNetwork hands the vectors to libnerfsig (mlp_pack_weights, field_*)."""
import math

import torch
import torch.nn as nn


def _pad16(n):
    return (n + 15) // 16 * 16


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        cfg = dict(network_config)
        if cfg.get("otype") != "FullyFusedMLP" or cfg.get("activation") != "ReLU" or cfg.get("output_activation") != "None":
            raise NotImplementedError("only FullyFusedMLP / ReLU / no output activation (the reference's configuration)")
        w, n_hidden = int(cfg["n_neurons"]), int(cfg["n_hidden_layers"])
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.widths = ((w, _pad16(n_input_dims)),) + ((w, w),) * (n_hidden - 1) + ((_pad16(n_output_dims), w),)
        g = torch.Generator().manual_seed(seed)
        chunks = []
        for fan_out, fan_in in self.widths:  # Xavier-uniform, as tiny-cuda-nn initialises FullyFusedMLP
            a = math.sqrt(6.0 / (fan_in + fan_out))
            chunks.append((torch.rand(fan_out * fan_in, generator=g) * 2 - 1) * a)
        self.params = nn.Parameter(torch.cat(chunks))

    def forward(self, x):
        raise RuntimeError("tcnn_compat.Network only stores parameters; NeRFNetwork evaluates it through libnerfsig")


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config):
        super().__init__()
        cfg = dict(encoding_config)
        if cfg.get("otype") != "SphericalHarmonics" or int(cfg.get("degree", 0)) != 4 or n_input_dims != 3:
            raise NotImplementedError("only the degree-4 SphericalHarmonics encoding of 3-vectors (the reference's configuration)")
        self.n_input_dims, self.n_output_dims = 3, 16
        self.params = nn.Parameter(torch.zeros(0))

    def forward(self, x):
        raise RuntimeError("tcnn_compat.Encoding only mirrors the module surface; NeRFNetwork evaluates it through libnerfsig")
