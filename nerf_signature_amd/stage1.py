"""Stage-1 (clean model) training, SURVEY.md 8(f) N3: the same field kernels with every parameter trainable.

Mirror of /root/reference/nerf/network_hash.py (NeRFNetwork without codebook / decoder; get_params :154-166) and of the
loop body of the stage-1 trainer (/root/reference/nerf/utils.py:469-517 train_step, :852-869 loop with the density-grid
refresh every `update_extra_interval` steps).

Gradient flow (csrc/field.hip "stage-1", csrc/stage1_fused.hip, csrc/stage1.hip, csrc/hashgrid.hip "planned variant over the 16 base levels"):
field_fwd_trace saves each layer's input; field_bwd_wgrad back-propagates, keeps each layer's pre-activation gradient on the chip
and reduces the five weight gradients over the points on MFMA in the same launch, leaving the gradient of all 32 encoder features
(fused=False: field_bwd_trace writes the pre-activation gradients out and field_wgrad reads them back -- the two-launch route of
the first version, kept as the cross-check); the base-table gradients are one planned owner-computes scatter over the 16 levels
(hg_levels_plan beside the forward pass, hg_levels_scatter behind the backward) -- the counterpart of the reference's 16
embedding_dense_backward calls.

Three ways to drive it:
  * CleanNeRFNetwork under autograd (any caller, e.g. the reference's own stage-1 Trainer through render());
  * CleanLoop            the loop body, eager;
  * GraphedCleanLoop     the loop body as explicit kernel calls on static buffers, captured once into a hipGraph and replayed:
                         no autograd, no allocation, no host read inside a step; the density grid is refreshed between replays
                         every `update_extra_interval` steps as the reference does; data-parallel ranks exchange ONE flat
                         buffer [16 table gradients | MLP gradients] per step (as two collectives where the tables can then travel beside the weight-gradient reduction)."""
import contextlib
import ctypes

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _native as nv
from . import dp
from . import fieldops as fo
from . import tcnn_compat as tcnn
from .capture import SegmentedCapture
from .optim import _bump_versions
from .hash_encoding import HashEmbedder
from .raymarching import padded_point_count
from .renderer import NeRFRenderer

T_ROWS = fo.T_ROWS
N_SIGMA, N_COLOR = 3072, 7168      # sigma_net.params / color_net.params (tcnn layout)


RESOLUTIONS = (16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047)      # csrc/hashgrid.h kBaseResolution (hash_encoding.py:60,100 in fp32)
_LIVE_ROWS = {}


def live_rows(level):
    """The rows of base table `level` that can EVER carry a gradient, sorted: what the level's grid corners hash to (hash_encoding.py:11-22: x ^ y * 2654435761 ^
    z * 805459861, 19 bits; cell indices 0..res from a coordinate in [0, 1], upper corners one more).  A static bound, the same on every rank: the coarse levels
    have few corners -- 5832 at level 0 -- scattered by the hash over the table's 2^19 rows, and every other row's gradient is an exact zero the owners store."""
    if level not in _LIVE_ROWS:
        import numpy as np
        n = RESOLUTIONS[level] + 2
        if n ** 3 >= 4 * T_ROWS:
            rows = np.arange(T_ROWS, dtype=np.int64)        # (far more corners than rows: every row is hit)
        else:
            i = np.arange(n, dtype=np.uint32)
            h = (i[:, None, None] ^ (i[None, :, None] * np.uint32(2654435761)) ^ (i[None, None, :] * np.uint32(805459861))) & np.uint32(T_ROWS - 1)
            rows = np.unique(h.reshape(-1)).astype(np.int64)
        _LIVE_ROWS[level] = torch.from_numpy(rows)
    return _LIVE_ROWS[level]


SPARSE_EXCHANGE_LEVELS = tuple(l for l in range(16) if (RESOLUTIONS[l] + 2) ** 3 < T_ROWS // 2)      # levels 0..4: 5832 .. 216 000 corners for 524 288 rows


def _stride(M):
    return (M + 31) // 32 * 32


class _Traces:
    """The buffers one field pass of `M` points (capacity) leaves for its backward: planes, layer inputs, pre-activation gradients."""

    def __init__(self, M, dev, with_grads=True, fused=True, half=False):
        st = self.stride = _stride(M)
        self.fused = fused
        # half: the layer inputs kept as fp16 (448 instead of 896 bytes per point written by the forward and read back by the backward) -- what the reference's MLPs
        # keep for their backward (tinycudann FullyFusedMLP: fp16 activations); the one-launch backward only (field_fwd_trace_f16 / field_bwd_wgrad_f16)
        self.half = bool(half)
        if self.half and not fused:
            raise ValueError("fp16 traces go with the one-launch backward (fused=True): field_bwd_trace + field_wgrad read fp32 rows")
        f32 = dict(dtype=torch.float32, device=dev)
        self.M = M
        self.planes = torch.empty(17, st, 2, **f32)
        self.sig = torch.empty(M, **f32)
        self.rgb = torch.empty(M, 3, **f32)
        self.masks = torch.empty(st, fo.MASK_WORDS, dtype=torch.int32, device=dev)
        self.act = [torch.empty(w, st, dtype=torch.float16 if self.half else torch.float32, device=dev) for w in (64, 32, 64, 64)]               # hs, cin, h1, h2
        if with_grads:
            self.alloc_grads()

    def alloc_grads(self):
        f32 = dict(dtype=torch.float32, device=self.sig.device)
        st = self.stride
        self.d_planes = torch.empty(16, st, 2, **f32)
        if self.fused:      # field_bwd_wgrad: the pre-activation gradients never leave the chip
            self.d = None
            self.wgrad_scratch = torch.empty(int(nv.fn("field_bwd_wgrad_scratch_bytes")(self.M)), dtype=torch.uint8, device=self.sig.device)
            return
        self.d = [torch.empty(w, st, **f32) for w in (64, 16, 64, 64, 16)]              # d_hs, d_so, d_h1, d_h2, d_out
        self.wgrad_scratch = torch.empty(int(nv.fn("field_wgrad_scratch_bytes")(self.M)), dtype=torch.uint8, device=self.sig.device)


def _forward_trace(tr, xyzs, dirs, bound, base_ptrs, packed, rows=None):
    s = nv.stream()
    if tr.half:
        nv.call("hg_encode_planes" if rows is None else "hg_encode_planes_rows", nv.ptr(xyzs), tr.M, *(() if rows is None else (nv.ptr(rows),)), float(bound), base_ptrs, None,
                nv.ptr(tr.planes), s)
        nv.call("field_fwd_trace_f16", nv.ptr(xyzs), nv.ptr(dirs), tr.M, nv.ptr(rows), float(bound), base_ptrs, nv.ptr(packed), nv.ptr(tr.planes), nv.ptr(tr.sig),
                nv.ptr(tr.rgb), nv.ptr(tr.masks), *[nv.ptr(a) for a in tr.act], s)
    elif rows is None:
        nv.call("hg_encode_planes", nv.ptr(xyzs), tr.M, float(bound), base_ptrs, None, nv.ptr(tr.planes), s)
        nv.call("field_fwd_trace", nv.ptr(xyzs), nv.ptr(dirs), tr.M, float(bound), base_ptrs, nv.ptr(packed), nv.ptr(tr.planes), nv.ptr(tr.sig),
                nv.ptr(tr.rgb), nv.ptr(tr.masks), *[nv.ptr(a) for a in tr.act], s)
    else:
        nv.call("hg_encode_planes_rows", nv.ptr(xyzs), tr.M, nv.ptr(rows), float(bound), base_ptrs, None, nv.ptr(tr.planes), s)
        nv.call("field_fwd_trace_rows", nv.ptr(xyzs), nv.ptr(dirs), tr.M, nv.ptr(rows), float(bound), base_ptrs, nv.ptr(packed), nv.ptr(tr.planes),
                nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(tr.masks), *[nv.ptr(a) for a in tr.act], s)


def _backward_trace(tr, g_sigma, g_rgb, packed, g_sigma_params, g_color_params, rows=None, wgrad_stream=None):
    """MLP backward + the weight gradients (written, not accumulated into); leaves d_planes for the level scatter.  One launch (tr.fused), or
    field_bwd_trace + field_wgrad.  wgrad_stream (two-launch route only): the weight-gradient reduction is issued there (the caller joins it): it
    and the table scatter both need the backward's traces and nothing of each other."""
    s = nv.stream()
    if tr.fused:
        nv.call("field_bwd_wgrad_f16" if tr.half else "field_bwd_wgrad", tr.M, nv.ptr(rows), nv.ptr(g_sigma), nv.ptr(g_rgb), nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(tr.masks), nv.ptr(packed),
                nv.ptr(tr.planes), *[nv.ptr(a) for a in tr.act], nv.ptr(tr.d_planes), nv.ptr(tr.wgrad_scratch), nv.ptr(g_sigma_params), nv.ptr(g_color_params), s)
        return
    if rows is None:
        nv.call("field_bwd_trace", tr.M, nv.ptr(g_sigma), nv.ptr(g_rgb), nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(tr.masks), nv.ptr(packed),
                *[nv.ptr(t) for t in tr.d], nv.ptr(tr.d_planes), s)
    else:
        nv.call("field_bwd_trace_rows", tr.M, nv.ptr(rows), nv.ptr(g_sigma), nv.ptr(g_rgb), nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(tr.masks),
                nv.ptr(packed), *[nv.ptr(t) for t in tr.d], nv.ptr(tr.d_planes), s)
    if wgrad_stream is not None:
        wgrad_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(wgrad_stream) if wgrad_stream is not None else contextlib.nullcontext():
        nv.call("field_wgrad", tr.M, nv.ptr(rows), nv.ptr(tr.planes), *[nv.ptr(a) for a in tr.act], *[nv.ptr(t) for t in tr.d], nv.ptr(tr.wgrad_scratch),
                nv.ptr(g_sigma_params), nv.ptr(g_color_params), nv.stream())


class _CleanFieldFunction(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, xyzs, dirs, bound, sigma_params, color_params, *base):
        xyzs, dirs = xyzs.contiguous(), dirs.contiguous()
        M, dev = xyzs.shape[0], xyzs.device
        packed = fo.pack_weights(sigma_params, color_params)          # the weights change every step
        base_ptrs = nv.ptr_array([fo._check_table(t.detach(), "base table") for t in base])
        tr = _Traces(M, dev, with_grads=False)
        _forward_trace(tr, xyzs, dirs, bound, base_ptrs, packed)
        ctx.table_grads = [t.requires_grad for t in base]
        ctx.plan = None
        if M and all(ctx.table_grads):      # where every (point, level) entry of the table scatter will go: positions only
            ctx.plan = torch.empty(int(nv.fn("hg_levels_plan_bytes")(M)), dtype=torch.uint8, device=dev)
            nv.call("hg_levels_plan", nv.ptr(xyzs), M, None, float(bound), nv.ptr(ctx.plan), nv.stream())
        sig, rgb = tr.sig, tr.rgb
        ctx.save_for_backward(sig, rgb)      # (outputs: kept through autograd's own mechanism, not as attributes of ctx)
        tr.sig = tr.rgb = None
        ctx.tr, ctx.packed, ctx.xyzs = tr, packed, xyzs
        ctx.bound, ctx.M = float(bound), M
        return sig, rgb

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_sigma, g_rgb):
        tr, xyzs, M, dev = ctx.tr, ctx.xyzs, ctx.M, ctx.xyzs.device
        tr.sig, tr.rgb = ctx.saved_tensors
        g_sp = torch.empty(N_SIGMA, dtype=torch.float32, device=dev)
        g_cp = torch.empty(N_COLOR, dtype=torch.float32, device=dev)
        if M == 0:
            return (None, None, None, g_sp.zero_(), g_cp.zero_()) + tuple(torch.zeros(T_ROWS, 2, dtype=torch.float32, device=dev) if need else None
                                                                         for need in ctx.table_grads)
        tr.alloc_grads()
        _backward_trace(tr, g_sigma.contiguous().float(), g_rgb.contiguous().float(), ctx.packed, g_sp, g_cp)
        # base-table gradients: owner-computes scatter, all 16 levels at once (every row written by its owner: no zero fill)
        if ctx.plan is not None:
            tables = torch.empty(16, T_ROWS, 2, dtype=torch.float32, device=dev)
            nv.call("hg_levels_scatter", nv.ptr(xyzs), M, None, ctx.bound, nv.ptr(tr.d_planes), tr.stride, nv.ptr(ctx.plan),
                    nv.ptr_array([tables[l] for l in range(16)]), nv.stream())
            grads = list(tables.unbind(0))
        else:
            grads = []
            for level, need in enumerate(ctx.table_grads):
                if not need:
                    grads.append(None)
                    continue
                G = torch.zeros(T_ROWS, 2, dtype=torch.float32, device=dev)
                nv.call("hg_scatter_level", nv.ptr(xyzs), ctx.bound, nv.ptr(tr.d_planes[level]), M, level, nv.ptr(G), nv.stream())
                grads.append(G)
        ctx.tr = ctx.plan = None
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

    def trainable(self):
        """The 18 tensors a step updates, in the order of the flat gradient buffer: 16 base tables, sigma MLP, colour MLP."""
        return list(self.encoder.tables()) + [self.sigma_net.params, self.color_net.params]


def train_step(model, data, render_kwargs):
    """nerf/utils.py:469-517 for RGB ground truth: render with perturbed samples, MSE.  Returns (pred_rgb, loss)."""
    images = data["images"]
    if images.shape[-1] != 3:
        raise NotImplementedError("stage-1 train_step: RGB ground truth (the RGBA branch blends with a random background, utils.py:489-498)")
    out = model.render(data["rays_o"], data["rays_d"], None, staged=False, bg_color=1, perturb=data.get("perturb", True),
                       force_all_rays=data.get("force_all_rays", False), **render_kwargs)
    loss = ((out["image"] - images) ** 2).mean(-1).mean()
    return out["image"], loss


class CleanLoop:
    """Loop body of the stage-1 trainer: render a batch of rays, MSE against the images, backward, optimiser step; every
    `update_extra_interval` steps refresh the density grid (utils.py:852-869).  With more than one rank the gradients are averaged
    (dp.allreduce_gradients) before the optimiser step."""

    def __init__(self, model, optimizer, render_kwargs, update_extra_interval=16, lr_scheduler=None, ema_decay=None):
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.render_kwargs = dict(render_kwargs)
        self.update_extra_interval = update_extra_interval
        self.global_step = 0
        # torch_ema.ExponentialMovingAverage(model.parameters(), decay) as utils.py:389-390 builds it (main_nerf.py:130: 0.95), restated with tensor operators
        self.ema_decay = ema_decay
        self.ema_shadow = None if ema_decay is None else [p.detach().clone() for p in model.trainable()]
        self.ema_updates = 0

    @torch.no_grad()
    def _ema_update(self):
        self.ema_updates += 1
        decay = min(self.ema_decay, (1 + self.ema_updates) / (10 + self.ema_updates))
        for s_, p in zip(self.ema_shadow, self.model.trainable()):
            tmp = s_ - p
            tmp.mul_(1.0 - decay)
            s_.sub_(tmp)

    def step(self, data):
        if self.model.cuda_ray and self.global_step % self.update_extra_interval == 0:
            self.model.update_extra_state()
        self.global_step += 1
        self.optimizer.zero_grad(set_to_none=True)
        image, loss = train_step(self.model, data, self.render_kwargs)
        loss.backward()
        if dp.world_size() > 1:
            dp.allreduce_gradients([p for g in self.optimizer.param_groups for p in g["params"]])
        self.optimizer.step()
        if self.ema_shadow is not None:      # utils.py:761-762: after the optimiser step
            self._ema_update()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return image, loss


class GraphedCleanLoop:
    """The stage-1 loop body captured into a hipGraph (module docstring).

    One step = [draw rays (optional device-side loader)] -> capacity march with perturbed starts -> plan of the table scatter (side stream) |
    encoder -> MLPs with saved layer inputs -> compositing + background -> MSE and its gradient -> compositing backward -> MLP backward ->
    weight gradients -> table scatter -> [all-reduce of the flat gradient buffer] -> Adam over 16 tables + both MLPs -> re-pack of the MLP
    weights for the next step.  Between replays, every `update_extra_interval` steps: NeRFRenderer.update_extra_state (eager; it reads the
    sample totals of the last 16 steps back, the one host read of the loop) and a capacity check.

    capacity: rows of the point buffers.  Every kernel walks only the rows the march produced (a device count), so a generous capacity
    costs memory, not time; a step that produced more points than fit dropped its overflowing rays, exactly like the reference's bounded
    march (raymarching.cu:416 with M = mean_count, the default of its stage-1 train_step: force_all_rays=False) -- `overflowed()` reports it,
    and the grid-refresh check grows the buffers and captures again when a step came within 10 % of them.
    sampler (rays.DeviceRaySampler): draws pose, pixels, rays and ground truth inside the graph from the step count; otherwise call
    step(data) with 'rays_o', 'rays_d' [..,3] and 'images' [..,3] of `n_rays` rays (copied into the static buffers)."""

    LOSS_RING = 1024
    PLAN_OVERLAP_MIN_ROWS = 600_000      # overlap_plan="auto": buffer capacity (rows) from which the scatter plan runs on its own stream (~400 k points per step)

    def __init__(self, model, optimizer, render_kwargs, n_rays, sampler=None, update_extra_interval=16, lr_lambda=None, headroom=0.5, perturb=True,
                 capacity=None, overlap_plan="auto", capture=True, seed=0, fused_backward=True, fused_composite=True, fused_table_adam=None, sparse_exchange=True,
                 device_refresh=True, trace_dtype="f32", ema_decay=None):
        if not model.cuda_ray:
            raise ValueError("GraphedCleanLoop drives the occupancy-grid path (cuda_ray=True)")
        if model.density_scale != 1:
            raise NotImplementedError("GraphedCleanLoop: density_scale != 1")
        if not isinstance(optimizer, torch.optim.Adam) or any(g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False)
                                                               for g in optimizer.param_groups):
            raise TypeError("GraphedCleanLoop steps a plain torch.optim.Adam (the reference's, main_nerf.py:122) through opt_adam_dense")
        self.model, self.optimizer = model, optimizer
        self.render_kwargs = dict(render_kwargs)
        self.dt_gamma, self.max_steps = float(render_kwargs.get("dt_gamma", 0)), int(render_kwargs.get("max_steps", 1024))
        self.T_thresh = float(render_kwargs.get("T_thresh", 1e-4))
        self.n_rays, self.sampler, self.perturb = int(n_rays), sampler, bool(perturb)
        self.update_extra_interval = int(update_extra_interval)
        self.lr_lambda, self.headroom, self.capacity = lr_lambda, float(headroom), capacity
        dev = self.device = model.density_bitfield.device
        f32 = dict(dtype=torch.float32, device=dev)
        N = self.n_rays
        self.rays_o, self.rays_d, self.gt = (torch.zeros(N, 3, **f32) for _ in range(3))
        self.bg = torch.ones(3, **f32)
        self.ws, self.depth, self.depth_out = (torch.empty(N, **f32) for _ in range(3))
        self.image, self.image_out, self.g_image, self._g_image_books = (torch.empty(N, 3, **f32) for _ in range(4))      # (_g_image_books: what clean_loss writes when
        # it only keeps the books -- the same values as g_image)
        self.loss = torch.zeros(1, **f32)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)          # steps so far (advanced by the loss kernel)
        self.count_ring = torch.zeros(16, 2, dtype=torch.int32, device=dev)    # the march's (points, rays) of the last 16 steps
        self.loss_ring = torch.zeros(self.LOSS_RING, **f32)
        # perturb=True: the per-ray start offsets of a step (raymarching.py:213 `torch.rand(N)`): the first step's from torch's generator, every later
        # step's written by the previous step's loss kernel as a function of (seed, step, ray)
        self.seed = int(seed)
        self.noises = torch.rand(N, **f32) if self.perturb else None
        # one flat gradient buffer: [16 tables | sigma MLP | colour MLP] -- the scatter owners and the weight-gradient reduction WRITE it
        # (no zero fill), a data-parallel step all-reduces it in one collective, Adam reads it
        self.params = model.trainable()
        # Data-parallel ranks exchange only what can be non-zero: the coarse levels' gradients live on the few rows their grid corners hash to
        # (live_rows: a static set, identical on every rank), so those levels travel as their live rows, packed behind the dense levels -- 16 tables = 64 MiB
        # become levels 5..15 dense + 347 k rows = 46.6 MiB; the sums are the dense exchange's bit for bit (the same two operands per element).  Layout of the one
        # buffer: [levels 0..4 dense (local) | levels 5..15 dense | packed live rows of levels 0..4 | sigma MLP | colour MLP]; exchanged: everything behind levels 0..4.
        self.sparse_exchange = bool(sparse_exchange) and dp.exchange_active()
        n_tab = 16 * T_ROWS * 2
        self._n_local = len(SPARSE_EXCHANGE_LEVELS) * T_ROWS * 2 if self.sparse_exchange else 0
        assert SPARSE_EXCHANGE_LEVELS == tuple(range(len(SPARSE_EXCHANGE_LEVELS)))
        self._live_index = (torch.cat([live_rows(l) + l * T_ROWS for l in SPARSE_EXCHANGE_LEVELS]).to(dev) if self.sparse_exchange else None)
        n_packed = 2 * self._live_index.numel() if self.sparse_exchange else 0
        self.flat = torch.empty(n_tab + n_packed + N_SIGMA + N_COLOR, **f32)
        self.g_tables = self.flat[:n_tab].view(16, T_ROWS, 2)
        self.g_packed = self.flat[n_tab:n_tab + n_packed].view(-1, 2)
        self.g_sigma = self.flat[n_tab + n_packed:n_tab + n_packed + N_SIGMA]
        self.g_color = self.flat[n_tab + n_packed + N_SIGMA:]
        self.packed = torch.empty(int(nv.fn("mlp_packed_bytes")()), dtype=torch.uint8, device=dev)
        self._adam_scratch = [torch.empty(64, **f32), torch.empty(64, **f32)]
        if any(g["betas"] != optimizer.param_groups[0]["betas"] or g["eps"] != optimizer.param_groups[0]["eps"] for g in optimizer.param_groups):
            raise NotImplementedError("GraphedCleanLoop expects one (betas, eps) for all parameter groups (the reference's, main_nerf.py:122)")
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.lr_dev = torch.tensor(self.base_lr, **f32)
        # the scatter plan beside the encoder, on a stream of its own: a fork and a join in the captured graph, ~30 us on this runtime -- more than the whole plan
        # takes on a trained scene's sparse grid.  "auto": forked from PLAN_OVERLAP_MIN_ROWS buffer rows on (decided at every capture); same box, two rounds:
        # 673 k points 1.102 / 1.098 ms forked against 1.141 / 1.137 serial, 125 k points 0.376 / 0.373 forked against 0.365 / 0.367 serial
        self.overlap_plan = overlap_plan
        self.plan_stream = torch.cuda.Stream() if overlap_plan is True else None
        self.graph, self.tr, self.rec, self.plan = None, None, None, None
        self.capture = bool(capture)      # False: the same explicit kernel sequence issued eagerly every step (tests, debugging)
        self.fused_composite = bool(fused_composite)    # False: compositing forward, clean_loss, compositing backward as three launches in a row
        self.fused_backward = bool(fused_backward)      # False: field_bwd_trace + field_wgrad (the latter on the plan's stream) instead of field_bwd_wgrad
        if trace_dtype not in ("f32", "f16"):
            raise ValueError("trace_dtype: 'f32' (strict: the saved layer inputs as the forward computed them) | 'f16' (half the trace bytes; the reference's precision)")
        self.trace_half = trace_dtype == "f16"
        # the tables' Adam step inside the scatter's owners (hg_levels_scatter_adam): their gradient never leaves the chip -- no 64 MiB written and read back, no
        # table pass on the step's tail; the tables' .grad is then NOT produced.  None: on unless the gradients are exchanged between ranks (which needs them in memory)
        self.fused_table_adam = (not dp.exchange_active()) if fused_table_adam is None else bool(fused_table_adam)
        if self.fused_table_adam and dp.exchange_active():
            raise ValueError("GraphedCleanLoop: fused_table_adam steps the tables before any exchange -- one process only")
        # the grid refresh as device-side work, replayed as a graph of its own between the step's replays (gridrefresh.DeviceGridRefresh: no host read);
        # False: NeRFRenderer.update_extra_state, the reference's form (three host synchronisations per refresh)
        self.device_refresh = bool(device_refresh)
        self._refresh = None
        # the trainer's moving average of the parameters (main_nerf.py:130 ema_decay=0.95; torch_ema semantics, csrc/stage1.hip opt_ema_update): one launch at the
        # step's tail, inside the graph; read it through ema_parameters() / the ema_weights() context (utils.py:801-811 evaluates and checkpoints with it)
        self.ema_decay = None if ema_decay is None else float(ema_decay)
        self.ema_shadow = None if ema_decay is None else [p.detach().clone() for p in self.params]
        self._peak_host = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == "cuda" else None
        self._peak_ready = None      # event behind the copy of the last window's peak into _peak_host
        self.global_step = 0
        self.recaptures = 0
        self.bytes_exchanged_per_step = (self.flat.numel() - self._n_local) * 4 if dp.exchange_active() else 0

    # ---- pieces of one step (run eagerly once as warm-up, then under capture)
    def _march(self):
        return self.model.march_ahead(self.rays_o, self.rays_d, self.dt_gamma, self.max_steps, perturb=self.perturb, capacity=self.capacity,
                                      noises=self.noises if self.perturb else None)

    def _forward_backward(self):
        m, tr = self.model, self.tr
        self._wg_joined = False
        if self.sampler is not None:
            self.sampler.sample_into(self.step_dev, self.rays_o, self.rays_d, self.gt)
        rec = self.rec = self._march()
        rows = rec["counter"]                                     # [points, rays] int32: element 0 is the device row count
        xyzs, dirs, M, N = rec["xyzs"], rec["dirs"], self.capacity, self.n_rays
        main = torch.cuda.current_stream()
        if self.plan_stream is not None:
            self.plan_stream.wait_stream(main)
            with torch.cuda.stream(self.plan_stream):
                nv.call("hg_levels_plan", nv.ptr(xyzs), M, nv.ptr(rows), float(m.bound), nv.ptr(self.plan), nv.stream())
                self._plan_done = torch.cuda.Event()
                self._plan_done.record()
        else:
            nv.call("hg_levels_plan", nv.ptr(xyzs), M, nv.ptr(rows), float(m.bound), nv.ptr(self.plan), nv.stream())
        base_ptrs = nv.ptr_array([t.detach() for t in m.encoder.tables()])
        _forward_trace(tr, xyzs, dirs, m.bound, base_ptrs, self.packed, rows=rows)
        s = nv.stream()
        books = (nv.ptr(self.step_dev), nv.ptr(rows), nv.ptr(self.count_ring), nv.ptr(self.loss_ring), self.LOSS_RING, nv.ptr(self.noises), N, self.seed)
        if self.fused_composite:
            # compositing + background, the MSE's gradient and the compositing backward in ONE launch (a wave owns a ray in all three); the loss VALUE and the loop's
            # books (sample totals, loss ring, the next step's march offsets, the step count) remain clean_loss's, behind it on the same stream -- two launches in a
            # row instead of three (same box: dense -1.7 %, sparse grid -1.3 %; with clean_loss beside the MLP backward on the plan's stream the extra fork / join of
            # the graph cost more than the launch saved: +3 %)
            nv.call("rm_composite_train_mse", nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(rec["deltas"]), nv.ptr(rec["rays"]), M, N, self.T_thresh, nv.ptr(rec["nears"]),
                    nv.ptr(rec["fars"]), nv.ptr(self.bg), 0, nv.ptr(self.gt), 3 * N, 1.0 / dp.world_size(), nv.ptr(self.ws), nv.ptr(self.depth), nv.ptr(self.image),
                    nv.ptr(self.image_out), nv.ptr(self.depth_out), nv.ptr(self.g_image), nv.ptr(self.g_sig), nv.ptr(self.g_rgb), s)
            nv.call("clean_loss", nv.ptr(self.image_out), nv.ptr(self.gt), 3 * N, 1.0 / dp.world_size(), nv.ptr(self.loss), nv.ptr(self._g_image_books), *books, s)
        else:
            nv.call("rm_composite_train_finish_fwd", nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(rec["deltas"]), nv.ptr(rec["rays"]), M, N, self.T_thresh,
                    nv.ptr(rec["nears"]), nv.ptr(rec["fars"]), nv.ptr(self.bg), 0, nv.ptr(self.ws), nv.ptr(self.depth), nv.ptr(self.image),
                    nv.ptr(self.image_out), nv.ptr(self.depth_out), s)
            # the loss of the global batch is the mean over the ranks' losses: each rank seeds 1 / world, the exchange sums
            nv.call("clean_loss", nv.ptr(self.image_out), nv.ptr(self.gt), 3 * N, 1.0 / dp.world_size(), nv.ptr(self.loss), nv.ptr(self.g_image), *books, s)
            nv.call("rm_composite_train_finish_bwd", None, nv.ptr(self.g_image), nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(rec["deltas"]), nv.ptr(rec["rays"]),
                    nv.ptr(self.ws), nv.ptr(self.image), nv.ptr(self.bg), 0, M, N, self.T_thresh, 1, nv.ptr(self.g_sig), nv.ptr(self.g_rgb), s)
        # two-launch route: the weight gradients (a streaming reduction) run beside the table scatter (store- and LDS-bound) on the plan's stream,
        # which has long finished the plan by then (stream order: plan, then the weight gradients); fused: they are done when the backward is
        _backward_trace(tr, self.g_sig, self.g_rgb, self.packed, self.g_sigma, self.g_color, rows=rows, wgrad_stream=self.plan_stream)
        if self.plan_stream is not None:      # the scatter needs the plan: an event recorded behind the plan, not the whole side stream
            main.wait_event(self._plan_done)
        if self.fused_table_adam:
            opt, tabs = self.optimizer, self.params[:16]
            group = opt.param_groups[0]
            self._ensure_state(tabs)
            nv.call("hg_levels_scatter_adam", nv.ptr(xyzs), M, nv.ptr(rows), float(m.bound), nv.ptr(tr.d_planes), tr.stride, nv.ptr(self.plan),
                    nv.ptr_array([p.data for p in tabs]), nv.ptr_array([opt.state[p]["exp_avg"] for p in tabs]),
                    nv.ptr_array([opt.state[p]["exp_avg_sq"] for p in tabs]), nv.ptr_array([opt.state[p]["step"] for p in tabs]), nv.ptr(self.lr_dev),
                    float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]), 1.0, nv.ptr(self._adam_scratch[0]), s)
            _bump_versions(tabs)
        else:
            nv.call("hg_levels_scatter", nv.ptr(xyzs), M, nv.ptr(rows), float(m.bound), nv.ptr(tr.d_planes), tr.stride, nv.ptr(self.plan),
                    nv.ptr_array([self.g_tables[l] for l in range(16)]), s)

    def _join_weight_gradients(self):
        """Once per step: a second wait in a LATER captured segment (the optimiser behind a collective that ended the segment in which the plan's stream was
        forked) would make the capturing stream wait on an event recorded outside its capture (stage1.py:390)."""
        if self.plan_stream is not None and not self._wg_joined:
            torch.cuda.current_stream().wait_stream(self.plan_stream)
        self._wg_joined = True

    def _exchange(self):
        """The data-parallel exchange of the flat gradient buffer (SUM; every rank seeded its loss with 1 / world).  Where a collective does not end a captured segment
        (eager, or captured inside the graph) it is issued in two pieces: the 16 table gradients (64 MiB) right behind the scatter -- they travel while the weight-gradient
        reduction still runs on the plan's stream, and the tables' Adam pass can follow them -- and the MLP gradients (40 KB) once that reduction has joined.  Between captured
        segments every forked stream has to join in front of a collective anyway: one buffer, one collective."""
        if not dp.exchange_active():
            return
        import torch.distributed as dist
        n_tab = 16 * T_ROWS * 2
        if self.sparse_exchange:      # the coarse levels' live rows, packed behind the dense levels (one gather; one scatter back behind the collective)
            coarse = self.flat[:self._n_local].view(-1, 2)
            torch.index_select(coarse, 0, self._live_index, out=self.g_packed)
        if dp.collective_ends_segment():
            flat = self.flat[self._n_local:]
            self._join_weight_gradients()           # one buffer, one collective: everything in it has to be there
            dp.collective(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM), name="all_reduce_stage1_gradients")
        else:
            tables, mlp = self.flat[self._n_local:n_tab + self.g_packed.numel()], self.flat[n_tab + self.g_packed.numel():]
            dp.collective(lambda: dist.all_reduce(tables, op=dist.ReduceOp.SUM), name="all_reduce_stage1_tables")
            self._join_weight_gradients()
            dp.collective(lambda: dist.all_reduce(mlp, op=dist.ReduceOp.SUM), name="all_reduce_stage1_mlp")
        if self.sparse_exchange:
            coarse.index_copy_(0, self._live_index, self.g_packed)

    def _ensure_state(self, params):
        """torch.optim.Adam's state of `params` in its capturable format (device step counts), created on first use."""
        opt = self.optimizer
        for p in params:
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not st["step"].is_cuda:
                st["step"] = st["step"].to(p.device)

    def _adam(self, params):
        """torch.optim.Adam's update of `params` from their `.grad` views through opt_adam_dense (state in torch's capturable format: device step counts)."""
        opt = self.optimizer
        group = opt.param_groups[0]
        self._ensure_state(params)
        n = len(params)
        numel = (ctypes.c_uint32 * n)(*[p.numel() for p in params])
        nv.call("opt_adam_dense", n, nv.ptr_array([p.data for p in params]), nv.ptr_array([p.grad for p in params]),
                nv.ptr_array([opt.state[p]["exp_avg"] for p in params]), nv.ptr_array([opt.state[p]["exp_avg_sq"] for p in params]),
                nv.ptr_array([opt.state[p]["step"] for p in params]), numel, nv.ptr(self.lr_dev), float(group["betas"][0]), float(group["betas"][1]),
                float(group["eps"]), 1.0, nv.ptr(self._adam_scratch[0 if params[0] is self.params[0] else 1]), nv.stream())
        _bump_versions(params)

    def _optimise(self):
        if self.fused_backward:                     # every gradient is there when the scatter is: one call (one prepare launch) over the 18 tensors
            self._join_weight_gradients()
            self._adam(self.params[16:] if self.fused_table_adam else self.params)      # (fused_table_adam: the owners have stepped the 16 tables)
        else:
            if not self.fused_table_adam:
                self._adam(self.params[:16])        # the tables need the scatter only: 448 MiB of streaming while the weight gradients finish
            self._join_weight_gradients()
            self._adam(self.params[16:])
        nv.call("mlp_pack_weights", nv.ptr(self.model.sigma_net.params.detach()), nv.ptr(self.model.color_net.params.detach()), nv.ptr(self.packed), nv.stream())
        if self.ema_shadow is not None:
            numel = (ctypes.c_uint32 * len(self.params))(*[p.numel() for p in self.params])
            nv.call("opt_ema_update", len(self.params), nv.ptr_array([p.data for p in self.params]), nv.ptr_array(self.ema_shadow), numel, nv.ptr(self.step_dev),
                    self.ema_decay, nv.stream())

    def ema_parameters(self):
        """The moving averages, in the order of model.trainable() (None without ema_decay)."""
        return self.ema_shadow

    @contextlib.contextmanager
    def ema_weights(self):
        """torch_ema's store() / copy_to() ... restore() around an evaluation or a checkpoint (utils.py:801-811): inside the block the model holds the averaged
        parameters (and the captured step's operand image of the MLP weights is rebuilt from them), afterwards the trained ones again."""
        if self.ema_shadow is None:
            yield
            return
        kept = [p.detach().clone() for p in self.params]
        repack = lambda: nv.call("mlp_pack_weights", nv.ptr(self.model.sigma_net.params.detach()), nv.ptr(self.model.color_net.params.detach()), nv.ptr(self.packed), nv.stream())
        with torch.no_grad():
            for p, s_ in zip(self.params, self.ema_shadow):
                p.copy_(s_)
            repack()
            _bump_versions(self.params)
        try:
            yield
        finally:
            with torch.no_grad():
                for p, k in zip(self.params, kept):
                    p.copy_(k)
                repack()
                _bump_versions(self.params)

    def _whole_step(self):
        self._forward_backward()
        self._exchange()
        self._optimise()
        return self.loss

    # ---- set-up
    def _size(self):
        """One synchronising march of the current rays: the point count the buffers have to hold (+ headroom)."""
        m = self.model
        if self.sampler is not None:
            self.sampler.sample_into(self.step_dev, self.rays_o, self.rays_d, self.gt)
        probe = m.march_ahead(self.rays_o, self.rays_d, self.dt_gamma, self.max_steps, perturb=False, capacity=128)
        n = int(probe["counter"][0])
        m.drop_marched()
        return padded_point_count(int(max(n, 4096) * (1.0 + self.headroom)))

    def _allocate(self):
        dev, M = self.device, self.capacity
        self.tr = _Traces(M, dev, fused=self.fused_backward, half=self.trace_half)
        self.g_sig = torch.empty(M, dtype=torch.float32, device=dev)
        self.g_rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
        self.plan = torch.empty(int(nv.fn("hg_levels_plan_bytes")(M)), dtype=torch.uint8, device=dev)

    @torch.no_grad()
    def prepare(self):
        if self.capacity is None:
            self.capacity = self._size()
        if self.overlap_plan == "auto":
            want = self.capacity >= self.PLAN_OVERLAP_MIN_ROWS
            if want != (self.plan_stream is not None):
                self.plan_stream = torch.cuda.Stream() if want else None
        self._allocate()
        for i, (p, g) in enumerate(zip(self.params, list(self.g_tables.unbind(0)) + [self.g_sigma, self.g_color])):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("GraphedCleanLoop: parameters must be contiguous float32 CUDA tensors")
            # views of the flat buffer: what the kernels write is what an optimiser / a checkpoint sees.  fused_table_adam: the tables' gradient exists inside the
            # scatter's owners only -- no stale view is left behind
            p.grad = None if (self.fused_table_adam and i < 16) else g.view_as(p)
        nv.call("mlp_pack_weights", nv.ptr(self.model.sigma_net.params.detach()), nv.ptr(self.model.color_net.params.detach()), nv.ptr(self.packed), nv.stream())
        # warm-up on a side stream (Adam state in its capturable format, module loading, RCCL's lazy set-up); it must not train
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._whole_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        self.graph = SegmentedCapture()
        if self.capture:
            self.graph.capture(self._whole_step)
        return self

    def _snapshot(self):
        state = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state[p].items()} for p in self.params if len(self.optimizer.state[p])}
        return ([p.detach().clone() for p in self.params], state, self.step_dev.clone(), self.count_ring.clone(), self.loss_ring.clone(),
                None if self.noises is None else self.noises.clone(), None if self.ema_shadow is None else [t.clone() for t in self.ema_shadow])

    def _restore(self, snap):
        values, state, step_dev, count_ring, loss_ring, noises, shadow = snap
        if shadow is not None:
            for t, v in zip(self.ema_shadow, shadow):
                t.copy_(v)
        for p, v in zip(self.params, values):
            p.copy_(v)
        for p in self.params:
            for k, v in self.optimizer.state[p].items():
                if torch.is_tensor(v):
                    if p in state and k in state[p]:
                        v.copy_(state[p][k].to(v.device))
                    else:
                        v.zero_()      # state created by the warm-up: back to its initial value, same storage (the graph holds its address)
        self.step_dev.copy_(step_dev)
        self.count_ring.copy_(count_ring)
        self.loss_ring.copy_(loss_ring)
        if noises is not None:
            self.noises.copy_(noises)
        nv.call("mlp_pack_weights", nv.ptr(self.model.sigma_net.params.detach()), nv.ptr(self.model.color_net.params.detach()), nv.ptr(self.packed), nv.stream())

    # ---- the loop
    @torch.no_grad()
    def refresh_grid(self):
        """update_extra_state between two replays (utils.py:852-857), with the loop's own ring of sample totals standing in for the renderer's."""
        m = self.model
        # the steps since the previous refresh (the reference resets local_step there, utils.py:852-858, renderer_wtmk.py:534-538), at most the ring's 16
        done = min(16, self.global_step, self.update_extra_interval if self.update_extra_interval > 0 else 16)
        if self.device_refresh:
            from .gridrefresh import DeviceGridRefresh
            if self._refresh is None:
                self._refresh = DeviceGridRefresh(m, seed=self.seed, capture=self.capture)
            if self.graph is None:      # (the first refresh comes before prepare(): the weights' operand image is not there yet)
                nv.call("mlp_pack_weights", nv.ptr(m.sigma_net.params.detach()), nv.ptr(m.color_net.params.detach()), nv.ptr(self.packed), nv.stream())
            self._refresh.run(self.packed, self.count_ring, self.step_dev, window=done)
        else:
            if done:      # their totals in step order into the renderer's ring, whose first `local_step` rows update_extra_state averages
                rows = torch.tensor([(self.global_step - done + i) % 16 for i in range(done)], device=self.count_ring.device)
                m.step_counter[:done].copy_(self.count_ring[rows])
                m.local_step = done
            m.update_extra_state()
        if self.graph is not None and done:
            peak = self._window_peak(min(16, self.global_step))
            if peak is not None and peak > 0.9 * self.capacity:
                self._grow(peak)

    def _window_peak(self, rows):
        """The largest sample total a step produced, over ALL ranks, as of the PREVIOUS refresh -- without synchronising: every refresh queues a copy of its
        window's maximum into pinned host memory and reads the one queued a refresh earlier (growth of the sample count is a trend over hundreds of steps; a step
        that does overflow in between drops its last rays like the reference's bounded march, and `overflowed()` reports it).
        All ranks: growing re-captures, and prepare()'s warm-up steps issue the step's collectives -- a rank that grew alone would run collectives its peers do
        not (ranks march different rays and cross the mark at different refreshes) and would end two optimiser steps apart from them.  The maximum over the ranks
        (one 4-byte all-reduce, queued with the copy) makes every rank take the same decision at the same refresh and arrive at the same capacity."""
        value = None
        if self._peak_ready is not None:
            self._peak_ready.synchronize()      # (queued 16 steps ago: long done)
            value = int(self._peak_host[0])
        cur = self.count_ring[:rows, 0].max().to(torch.int32).reshape(1)
        if dp.exchange_active() and dp.world_size() > 1:
            import torch.distributed as dist
            dist.all_reduce(cur, op=dist.ReduceOp.MAX)
        if self._peak_host is None:
            return int(cur)
        self._peak_host.copy_(cur, non_blocking=True)
        self._peak_ready = torch.cuda.Event()
        self._peak_ready.record()
        return value

    def _grow(self, peak):
        torch.cuda.synchronize()
        self.capacity = padded_point_count(int(peak * (1.0 + max(self.headroom, 0.25)) * 1.25))
        self.graph, self.rec = None, None
        self.model.drop_marched()
        self.recaptures += 1
        self.prepare()

    @torch.no_grad()
    def step(self, data=None):
        if self.update_extra_interval > 0 and self.global_step % self.update_extra_interval == 0:
            self.refresh_grid()
        if self.graph is None:
            if data is not None:
                self._set_batch(data)
            self.prepare()
        if data is not None:
            self._set_batch(data)
        if self.lr_lambda is not None:
            self.lr_dev.fill_(self.base_lr * self.lr_lambda(self.global_step))
        if self.capture:
            self.graph.replay()
        else:
            self._whole_step()
        self.global_step += 1
        return self.loss

    def _set_batch(self, data):
        if self.sampler is not None:
            raise ValueError("GraphedCleanLoop: this loop draws its own batches (sampler=)")
        for dst, key in ((self.rays_o, "rays_o"), (self.rays_d, "rays_d"), (self.gt, "images")):
            src = data[key]
            if src.numel() != dst.numel():
                raise ValueError(f"GraphedCleanLoop: '{key}' must hold {self.n_rays} x 3 values")
            dst.copy_(src.reshape(dst.shape), non_blocking=True)

    def overflowed(self):
        """True if one of the last (up to 16) steps produced more points than the buffers hold (one host read)."""
        done = min(16, self.global_step)
        return bool(done and int(self.count_ring[:done, 0].max()) > self.capacity)

    def losses(self, last=None):
        """The loss values of the last `last` steps (default: all that the ring still holds), oldest first (one host read)."""
        n = min(self.global_step, self.LOSS_RING if last is None else last)
        ring = self.loss_ring.cpu()
        return [float(ring[(self.global_step - n + i) % self.LOSS_RING]) for i in range(n)]

    def close(self):
        for p in self.params:
            p.grad = None
        self.model.drop_marched()
        self.graph = None
