"""Stage-1 (clean model) training support, SURVEY.md 8(f) N3: the same field kernels with every parameter trainable.

Mirror of /root/reference/nerf/network_hash.py (NeRFNetwork without codebook / decoder; get_params :154-166) and of the
loop body of the stage-1 trainer (/root/reference/nerf/utils.py:469-517 train_step, :852-869 density-grid refresh).

Gradient flow (csrc/field.hip "stage-1"): field_fwd_trace saves each layer's input, field_bwd_trace each layer's
pre-activation gradient and the gradient of all 32 encoder features.  The five weight gradients are reductions over the
point dimension of (pre-activation gradient) x (layer input)^T -- plain GEMMs, done by the BLAS library; the base-table
gradients are 16 owner-computes scatters (hg_scatter_level), the counterpart of the reference's 16
embedding_dense_backward calls."""
import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _native as nv
from . import fieldops as fo
from . import tcnn_compat as tcnn
from .hash_encoding import HashEmbedder
from .renderer import NeRFRenderer

T_ROWS = fo.T_ROWS


class _CleanFieldFunction(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, xyzs, dirs, bound, sigma_params, color_params, *base):
        xyzs, dirs = xyzs.contiguous(), dirs.contiguous()
        M, dev = xyzs.shape[0], xyzs.device
        stride = (M + 31) // 32 * 32
        packed = fo.pack_weights(sigma_params, color_params)          # the weights change every step
        base_ptrs = nv.ptr_array([fo._check_table(t.detach(), "base table") for t in base])
        planes = torch.empty(17, stride, 2, dtype=torch.float32, device=dev)
        s = nv.stream()
        nv.call("hg_encode_planes", nv.ptr(xyzs), M, float(bound), base_ptrs, None, nv.ptr(planes), s)
        sig = torch.empty(M, dtype=torch.float32, device=dev)
        rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
        masks = torch.empty(stride, fo.MASK_WORDS, dtype=torch.int32, device=dev)
        act = [torch.empty(w, stride, dtype=torch.float32, device=dev) for w in (64, 32, 64, 64)]     # hs, cin, h1, h2
        nv.call("field_fwd_trace", nv.ptr(xyzs), nv.ptr(dirs), M, float(bound), base_ptrs, nv.ptr(packed), nv.ptr(planes), nv.ptr(sig),
                nv.ptr(rgb), nv.ptr(masks), *[nv.ptr(a) for a in act], s)
        ctx.save_for_backward(xyzs, sig, rgb, masks, packed, planes, *act)
        ctx.bound, ctx.M = float(bound), M
        ctx.table_grads = [t.requires_grad for t in base]
        return sig, rgb

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_sigma, g_rgb):
        xyzs, sig, rgb, masks, packed, planes, a_hs, a_cin, a_h1, a_h2 = ctx.saved_tensors
        M, dev = ctx.M, xyzs.device
        stride = planes.shape[1]
        d_hs, d_h1, d_h2 = (torch.empty(64, stride, dtype=torch.float32, device=dev) for _ in range(3))
        d_so, d_out = (torch.empty(16, stride, dtype=torch.float32, device=dev) for _ in range(2))
        d_planes = torch.empty(16, stride, 2, dtype=torch.float32, device=dev)
        nv.call("field_bwd_trace", M, nv.ptr(g_sigma.contiguous().float()), nv.ptr(g_rgb.contiguous().float()), nv.ptr(sig), nv.ptr(rgb),
                nv.ptr(masks), nv.ptr(packed), nv.ptr(d_hs), nv.ptr(d_so), nv.ptr(d_h1), nv.ptr(d_h2), nv.ptr(d_out), nv.ptr(d_planes), nv.stream())
        # weight gradients: (pre-activation gradient) x (layer input)^T over the M points
        feat = planes[:16, :M]                                              # [16, M, 2] = the 32 features, level-major
        dW1s = torch.einsum("om,lmc->olc", d_hs[:, :M], feat).reshape(64, 32)
        dW2s = d_so[:, :M] @ a_hs[:, :M].t()
        dWc1 = d_h1[:, :M] @ a_cin[:, :M].t()
        dWc2 = d_h2[:, :M] @ a_h1[:, :M].t()
        dWc3 = d_out[:, :M] @ a_h2[:, :M].t()
        g_sp = torch.cat([dW1s.reshape(-1), dW2s.reshape(-1)])
        g_cp = torch.cat([dWc1.reshape(-1), dWc2.reshape(-1), dWc3.reshape(-1)])
        # base-table gradients: owner-computes scatter, all 16 levels in one launch (every row written by its owner: no zero fill)
        if all(ctx.table_grads):
            tables = torch.empty(16, T_ROWS, 2, dtype=torch.float32, device=dev)
            scratch = torch.empty(nv.fn("hg_scatter_levels_scratch_bytes")(M), dtype=torch.uint8, device=dev)
            nv.call("hg_scatter_levels", nv.ptr(xyzs), ctx.bound, nv.ptr(d_planes), M, stride, nv.ptr_array([tables[l] for l in range(16)]),
                    nv.ptr(scratch), nv.stream())
            grads = list(tables.unbind(0))
        else:
            grads = []
            for level, need in enumerate(ctx.table_grads):
                if not need:
                    grads.append(None)
                    continue
                G = torch.zeros(T_ROWS, 2, dtype=torch.float32, device=dev)
                nv.call("hg_scatter_level", nv.ptr(xyzs), ctx.bound, nv.ptr(d_planes[level]), M, level, nv.ptr(G), nv.stream())
                grads.append(G)
        return (None, None, None, g_sp, g_cp) + tuple(grads)


class CleanNeRFNetwork(NeRFRenderer):
    """nerf/network_hash.py NeRFNetwork: hash encoder + sigma MLP + SH + colour MLP, everything trainable."""

    def __init__(self, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        if (num_layers, hidden_dim, geo_feat_dim, num_layers_color, hidden_dim_color) != (2, 64, 15, 3, 64):
            raise NotImplementedError("the native field network implements the reference's default architecture")
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder = HashEmbedder(bounding_box=(0, 1), n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                                    base_resolution=16, finest_resolution=2048)
        mlp = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None"}
        self.sigma_net = tcnn.Network(32, 1 + geo_feat_dim, {**mlp, "n_neurons": hidden_dim, "n_hidden_layers": num_layers - 1}, seed=1337)
        self.encoder_dir = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4})
        self.in_dim_color = self.encoder_dir.n_output_dims + geo_feat_dim
        self.color_net = tcnn.Network(self.in_dim_color, 3, {**mlp, "n_neurons": hidden_dim_color, "n_hidden_layers": num_layers_color - 1}, seed=1338)

    def forward(self, x, d, message=None):
        if message is not None:
            raise ValueError("the clean model has no codebook")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _CleanFieldFunction.apply(x, d, self.bound, self.sigma_net.params, self.color_net.params, *self.encoder.tables())
        packed = fo.pack_weights(self.sigma_net.params, self.color_net.params)
        sigma, rgb, _, _ = fo.field_forward(x, d, self.bound, self.encoder.tables(), None, packed)
        return sigma, rgb

    def density(self, x, message=None):
        packed = fo.pack_weights(self.sigma_net.params, self.color_net.params)
        sigma, _, geo, _ = fo.field_forward(x, None, self.bound, self.encoder.tables(), None, packed, want_rgb=False, want_geo=True)
        return {"sigma": sigma, "geo_feat": geo}

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        packed = fo.pack_weights(self.sigma_net.params, self.color_net.params)
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=torch.float32, device=x.device)
            if mask.any():
                rgbs[mask] = fo.field_color(d[mask], geo_feat[mask], packed)
            return rgbs
        return fo.field_color(d, geo_feat, packed)

    def get_params(self, lr):
        return [{"params": self.encoder.parameters(), "lr": lr}, {"params": self.sigma_net.parameters(), "lr": lr},
                {"params": self.encoder_dir.parameters(), "lr": lr}, {"params": self.color_net.parameters(), "lr": lr}]


class CleanLoop:
    """Loop body of the stage-1 trainer: render a batch of rays, MSE against the images, backward, optimiser step; every
    `update_extra_interval` steps refresh the density grid (utils.py:852-869)."""

    def __init__(self, model, optimizer, render_kwargs, update_extra_interval=16, lr_scheduler=None):
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.render_kwargs = dict(render_kwargs)
        self.update_extra_interval = update_extra_interval
        self.global_step = 0

    def step(self, data):
        if self.model.cuda_ray and self.global_step % self.update_extra_interval == 0:
            self.model.update_extra_state()
        self.global_step += 1
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model.render(data["rays_o"], data["rays_d"], None, staged=False, bg_color=1, perturb=data.get("perturb", True),
                                force_all_rays=data.get("force_all_rays", False), **self.render_kwargs)
        loss = ((out["image"] - data["images"]) ** 2).mean(-1).mean()
        loss.backward()
        self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return out["image"], loss
