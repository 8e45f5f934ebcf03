"""Base multiresolution hash encoder + SH encoder (mirror of /root/reference/hash_encoding.py:48-196).

`HashEmbedder` keeps the reference's parameter surface -- an nn.ModuleList `embeddings` of 16
nn.Embedding(2^19, 2) -- so checkpoints load key-for-key, but evaluates through libnerfsig
(hg_encode_fwd): one kernel instead of ~25 tensor ops and 2 host syncs per level."""
import torch
import torch.nn as nn

from . import fieldops as fo


class HashEmbedder(nn.Module):
    def __init__(self, bounding_box, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                 finest_resolution=512):
        super().__init__()
        if (n_levels, n_features_per_level, log2_hashmap_size, base_resolution, finest_resolution) != (16, 2, 19, 16, 2048) \
                or tuple(float(b) for b in bounding_box) != (0.0, 1.0):
            raise NotImplementedError("the native encoder implements the configuration the reference instantiates "
                                      "(network_wtmk_tcnn.py:40-41): box (0,1), 16 levels x 2 features, T=2^19, 16..2048")
        self.bounding_box = bounding_box
        self.n_levels = n_levels
        self.n_features_per_level = n_features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = torch.tensor(base_resolution)
        self.finest_resolution = torch.tensor(finest_resolution)
        self.out_dim = n_levels * n_features_per_level
        self.b = torch.exp((torch.log(self.finest_resolution) - torch.log(self.base_resolution)) / (n_levels - 1))
        self.embeddings = nn.ModuleList([nn.Embedding(2 ** log2_hashmap_size, n_features_per_level) for _ in range(n_levels)])
        for emb in self.embeddings:
            nn.init.uniform_(emb.weight, a=-0.0001, b=0.0001)

    def tables(self):
        """The tables' Parameters, in level order (read straight from the sub-modules' parameter dicts: nn.Module attribute lookups are slow and the
        eager loops ask several times per render)."""
        return [m._parameters["weight"] for m in self.embeddings._modules.values()]

    def forward(self, x):
        """x: [B,3] in [0,1] -> [B,32]."""
        if torch.is_grad_enabled() and any(e.weight.requires_grad for e in self.embeddings):
            raise NotImplementedError("gradients of the base tables (stage-1 training) are outside the watermark path "
                                      "(SURVEY.md 8(f) N3); freeze them as network_wtmk_tcnn.py:90-95 does")
        return fo.encode(x, self.tables())


class SHEncoder(nn.Module):
    """Real spherical harmonics up to degree 4 with plain tensor ops (hash_encoding.py:114-196; degree 5 is not
    used by the path)."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3 and 1 <= degree <= 4
        self.input_dim, self.degree, self.out_dim = input_dim, degree, degree ** 2

    def forward(self, input, **kwargs):
        x, y, z = input.unbind(-1)
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols = [torch.full_like(x, 0.28209479177387814),
                -0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x,
                1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.31539156525252005 * (2.0 * zz - xx - yy),
                -1.0925484305920792 * xz, 0.5462742152960396 * (xx - yy),
                -0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z,
                -0.4570457994644658 * y * (4 * zz - xx - yy), 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
                -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy),
                -0.5900435899266435 * x * (xx - 3 * yy)]
        return torch.stack(cols[:self.out_dim], dim=-1)
