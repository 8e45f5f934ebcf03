"""Codebook ("message") hash encoder (mirror of /root/reference/hash_encoding_wtmk_bit.py:51-116).

Parameter surface as the reference: `embeddings` = 2*message_dim nn.Embedding(2^19, 2), bit i reading table
2i + message[i].  Evaluation uses the two identities documented in csrc/hashgrid.hip: the D selected tables
are pre-summed into one table S (hg_codebook_presum) and looked up once; the backward scatters one shared
gradient and fans it out to the D selected tables (unselected tables keep grad None, as in the reference)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import fieldops as fo


class _CodebookFunction(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x01, *selected):
        S = fo.codebook_presum(selected)
        out = fo.codebook_encode_literal(x01, [S])
        ctx.save_for_backward(x01)
        ctx.n = len(selected)
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x01,) = ctx.saved_tensors
        G = torch.zeros(fo.T_ROWS, 2, dtype=torch.float32, device=x01.device)
        fo.codebook_scatter(x01, g, G)
        slab = torch.empty(ctx.n, fo.T_ROWS, 2, dtype=torch.float32, device=x01.device)
        grads = [slab[i] for i in range(ctx.n)]
        fo.fanout_grad(G, grads)
        return (None,) + tuple(grads)


class HashEmbedder(nn.Module):
    def __init__(self, bounding_box, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                 finest_resolution=512, message_dim=16):
        super().__init__()
        if (n_features_per_level, log2_hashmap_size, base_resolution, finest_resolution) != (2, 19, 2048, 2048) \
                or n_levels != 2 * message_dim or tuple(float(b) for b in bounding_box) != (0.0, 1.0):
            raise NotImplementedError("the native codebook implements the configuration the reference instantiates "
                                      "(network_wtmk_tcnn.py:43-44): 2*message_dim tables, all at resolution 2048, T=2^19")
        if not 1 <= message_dim <= 64:
            raise ValueError("message_dim must be in [1, 64]")
        self.bounding_box = bounding_box
        self.n_levels = n_levels
        self.n_features_per_level = n_features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = torch.tensor(base_resolution)
        self.finest_resolution = torch.tensor(finest_resolution)
        self.out_dim = n_levels * n_features_per_level
        self.b = torch.exp((torch.log(self.finest_resolution) - torch.log(self.base_resolution)) / (n_levels - 1))
        self.message_dim = message_dim
        self.embeddings = nn.ModuleList([nn.Embedding(2 ** log2_hashmap_size, n_features_per_level) for _ in range(n_levels)])
        for emb in self.embeddings:
            nn.init.uniform_(emb.weight, a=-0.0001, b=0.0001)

    def tables(self):
        """The tables' Parameters, in level order (read straight from the sub-modules' parameter dicts: nn.Module attribute lookups are slow and the
        eager loops ask several times per render)."""
        return [m._parameters["weight"] for m in self.embeddings._modules.values()]

    def selected(self, message):
        bits = fo.message_bits(message)
        if len(bits) != self.message_dim:
            raise ValueError(f"message has {len(bits)} bits, encoder was built for {self.message_dim}")
        return fo.select_tables(self.tables(), bits)

    def forward(self, x, message):
        """x: [B,3] in [0,1], message: [message_dim] of 0./1. -> [B,2]."""
        return _CodebookFunction.apply(x, *self.selected(message))
