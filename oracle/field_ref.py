"""oracle/field_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Torch-CPU restatement of the Python half of the reference's watermarked render path:
the two hash encoders, the spherical-harmonics direction encoding, the two tiny
MLPs, trunc_exp, NeRFNetwork.forward, NeRFRenderer.run_cuda / run / render and
Trainer.train_step.  Each function cites the reference lines it follows
(paths relative to /root/reference).  The native half (marching/compositing) is
oracle/raymarch_ref.c, reached through oracle/raymarch_ref.py.

Pinning status
  * hash encoders, SH formulas, trunc_exp, decoder/loss arithmetic: PINNED -- checked in
    tests/test_oracle_golden.py against vectors produced by importing the real reference
    modules in the build container (tests/golden/make_golden.py, committed with its output).
  * MLP arithmetic and its parameter layout: PARITY UNPINNED.  The reference evaluates them
    with tinycudann (nerf/network_wtmk_tcnn.py:52-88), a third-party CUDA dependency that is
    neither vendored nor version-pinned by the reference and cannot run here.  What is
    restated is tiny-cuda-nn's documented FullyFusedMLP behaviour: no biases, ReLU hidden
    activations, weight matrices stored consecutively in one flat `params` vector, each
    [out, in] row-major with the first layer's input width and the last layer's output
    width padded to 16; inputs narrower than the padded width are padded with 1.0; the
    SphericalHarmonics encoding maps its [0,1] input back to [-1,1].  Arithmetic here is
    fp32 (tcnn computes in fp16), the stated tolerance of the path being 1e-3.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math

import numpy as np
import torch

from . import raymarch_ref as rm

LOG2_T = 19
HASH_MASK = (1 << LOG2_T) - 1
PRIME_Y, PRIME_Z = 2654435761, 805459861  # hash_encoding.py:16

# corner c = (c>>2 & 1, c>>1 & 1, c & 1) = (dx, dy, dz): hash_encoding.py:8
_CORNERS = torch.tensor([[(c >> 2) & 1, (c >> 1) & 1, c & 1] for c in range(8)], dtype=torch.int64)


def level_resolutions(n_levels=16, base_resolution=16, finest_resolution=2048):
    """Per-level resolution as the fp32 0-dim tensors the reference computes
    (hash_encoding.py:56-60,100): floor(base * b**i), b = exp((ln finest - ln base)/(L-1))."""
    base, fine = torch.tensor(base_resolution), torch.tensor(finest_resolution)
    b = torch.exp((torch.log(fine) - torch.log(base)) / (n_levels - 1))
    return [torch.floor(base * b ** i) for i in range(n_levels)]


def voxel_lookup(x01, resolution):
    """hash_encoding.py:24-46 for bounding box (0, 1): hashed row index of the 8 cell corners
    and the interpolation weights.  x01: [M,3] fp32.  Returns (rows [M,8] int64, w [M,3] fp32, cell [M,3] int32)."""
    xc = torch.clamp(x01, min=0, max=1)            # :33-35 (only indexing sees the clamp)
    cell_size = 1 / resolution                      # :37, fp32 0-dim
    cell = torch.floor(xc / cell_size).int()        # :39
    lo = cell * cell_size                           # :40
    hi = lo + torch.tensor([1.0, 1.0, 1.0]) * cell_size  # :41
    corners = cell.unsqueeze(1) + _CORNERS.unsqueeze(0)   # :43 -> int64 [M,8,3]
    rows = (corners[..., 0] ^ (corners[..., 1] * PRIME_Y) ^ (corners[..., 2] * PRIME_Z)) & HASH_MASK  # :11-22
    w = (x01 - lo) / (hi - lo)                      # :80 (unclamped x)
    return rows, w, cell


def trilerp(e, w):
    """hash_encoding.py:73-94.  e: [M,8,F] corner rows, w: [M,3]."""
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c00 = e[:, 0] * (1 - wx) + e[:, 4] * wx
    c01 = e[:, 1] * (1 - wx) + e[:, 5] * wx
    c10 = e[:, 2] * (1 - wx) + e[:, 6] * wx
    c11 = e[:, 3] * (1 - wx) + e[:, 7] * wx
    c0 = c00 * (1 - wy) + c10 * wy
    c1 = c01 * (1 - wy) + c11 * wy
    return c0 * (1 - wz) + c1 * wz


def base_encode(x01, tables):
    """HashEmbedder.forward, hash_encoding.py:96-111.  tables: 16 x [T,2] -> [M,32]."""
    outs = []
    for table, res in zip(tables, level_resolutions(len(tables), 16, 2048)):
        rows, w, _ = voxel_lookup(x01, res)
        outs.append(trilerp(table[rows], w))
    return torch.cat(outs, dim=-1)


def codebook_encode(x01, message, tables, faithful=False):
    """HashEmbedder(msg).forward, hash_encoding_wtmk_bit.py:99-116.  tables: 2D x [T,2];
    bit i reads table 2i + message[i]; every level has resolution 2048 (b == 1, :63); the D
    interpolated rows are summed.  faithful=True recomputes the (identical) lookup per bit as
    the reference does; the result is the same."""
    D = len(tables) // 2
    res = level_resolutions(2 * D, 2048, 2048)[0]  # network_wtmk_tcnn.py:43-44 -> b == 1, every level 2048
    outs = []
    rows = w = None
    for i in range(D):
        if faithful or rows is None:
            rows, w, _ = voxel_lookup(x01, res)
        outs.append(trilerp(tables[2 * i + int(message[i])][rows], w))
    return torch.sum(torch.stack(outs, dim=-1), dim=-1)


def sh4(d):
    """Degree-4 real spherical harmonics of d in [-1,1]^3, hash_encoding.py:157-183 -> [M,16]."""
    x, y, z = d.unbind(-1)
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    cols = [
        torch.full_like(x, 0.28209479177387814),
        -0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.31539156525252005 * (2.0 * zz - xx - yy),
        -1.0925484305920792 * xz, 0.5462742152960396 * (xx - yy),
        -0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z,
        -0.4570457994644658 * y * (4 * zz - xx - yy), 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
        -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy),
        -0.5900435899266435 * x * (xx - 3 * yy),
    ]
    return torch.stack(cols, dim=-1)


class _TruncExp(torch.autograd.Function):
    """activation.py:5-15."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def split_mlp_params(params, widths):
    """Flat tcnn `params` -> list of [out,in] matrices (layout assumption: module docstring)."""
    mats, off = [], 0
    for fan_out, fan_in in widths:
        mats.append(params[off:off + fan_out * fan_in].view(fan_out, fan_in))
        off += fan_out * fan_in
    assert off == params.numel()
    return mats


SIGMA_WIDTHS = ((64, 32), (16, 64))             # network_wtmk_tcnn.py:52-62  (32 -> 64 -> 1+15)
COLOR_WIDTHS = ((64, 32), (64, 64), (16, 64))   # network_wtmk_tcnn.py:78-88  (31 pad 32 -> 64 -> 64 -> 3 pad 16)


def round_operand(t, operands):
    """The MFMA operand rounding of the product's DEFAULT arithmetic ("f16": every matrix operand -- layer input and weight -- rounded once to fp16,
    products accumulated in fp32), as a straight-through op: the forward value is rounded, the gradient passes unchanged.  None: plain fp32."""
    if operands is None:
        return t
    assert operands in ("f16", "tcnn")
    return t + (t.half().float() - t).detach()


def _mlp_tcnn_forward(x, mats):
    """Forward VALUES of tiny-cuda-nn's FullyFusedMLP as published (tiny-cuda-nn src/fully_fused_mlp.cu; absent from /root/reference, restated): the input
    and the weights are __half, every layer's product is accumulated in __half fragments -- emulated as: the 16 products of one 16-wide k-step summed in fp32
    (a tensor-core instruction's internal precision), the running sum rounded to fp16 after every k-step -- hidden activations are stored as __half, the
    output is __half.  The reference reaches this through network_wtmk_tcnn.py:52-88 (tcnn.Network, "FullyFusedMLP")."""
    h = x.detach().half()
    for k, W in enumerate(mats):
        Wh = W.detach().half().float()
        acc = torch.zeros(h.shape[0], W.shape[0], dtype=torch.float16)
        hf = h.float()
        for k0 in range(0, W.shape[1], 16):
            acc = (acc.float() + hf[:, k0:k0 + 16] @ Wh[:, k0:k0 + 16].t()).half()
        h = torch.relu(acc) if k + 1 < len(mats) else acc
    return h.float()


def mlp(x, mats, operands=None):
    """operands="f16" emulates the kernels' default arithmetic (round_operand): the same values reach every ReLU up to the order of the fp32 accumulation,
    so the same side of every kink is taken -- the element-wise pin of the default mode (tests/test_gpu_field.py).
    operands="tcnn" emulates tiny-cuda-nn's (_mlp_tcnn_forward: fp16 operands, fp16 accumulate, fp16 activations and outputs); the VALUE is that emulation,
    the gradient the fp16-operand path's (straight-through): the mode exists to measure how far a tcnn-like evaluation sits from the product's
    (tests/test_gpu_field.py::test_distance_to_a_tcnn_like_evaluation), not to train with."""
    h = x
    for k, W in enumerate(mats):
        h = round_operand(h, operands) @ round_operand(W, operands).t()
        if k + 1 < len(mats):
            h = torch.relu(h)
    if operands == "tcnn":
        h = h + (_mlp_tcnn_forward(x, mats) - h).detach()
    return h


def density(x, message, P):
    """NeRFNetwork.density, network_wtmk_tcnn.py:126-144."""
    x01 = (x + P["bound"]) / (2 * P["bound"])
    feat = base_encode(x01, P["base_tables"])
    if message is not None:
        feat = torch.cat([feat[:, :-2], feat[:, -2:] + codebook_encode(x01, message, P["cb_tables"], P.get("faithful", False))], dim=-1)  # :106
    h = mlp(feat, split_mlp_params(P["sigma_params"], SIGMA_WIDTHS), P.get("mlp_operands"))
    return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}


def color(d, geo_feat, P):
    """NeRFNetwork.color without mask, network_wtmk_tcnn.py:147-176."""
    d01 = (d + 1) / 2
    enc = sh4(d01 * 2 - 1)
    cin = torch.cat([enc, geo_feat, torch.ones_like(enc[:, :1])], dim=-1)
    h = mlp(cin, split_mlp_params(P["color_params"], COLOR_WIDTHS), P.get("mlp_operands"))[:, :3]
    if P.get("mlp_operands") == "tcnn":      # network_wtmk_tcnn.py:174: torch.sigmoid on the network's __half output stays in fp16
        rgb = torch.sigmoid(h)
        return rgb + (torch.sigmoid(h.detach().half()).float() - rgb).detach()
    return torch.sigmoid(h)


def field_forward(x, d, message, P):
    """NeRFNetwork.forward, network_wtmk_tcnn.py:97-124 -> (sigma [M], rgb [M,3])."""
    dn = density(x, message, P)
    return dn["sigma"], color(d, dn["geo_feat"], P)


class _CompositeTrain(torch.autograd.Function):
    """raymarching.py:238-291 around the C oracle."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh):
        ws, depth, image = rm.composite_rays_train_forward(sigmas.detach().numpy(), rgbs.detach().numpy(), deltas.numpy(), rays.numpy(), T_thresh)
        ws, depth, image = torch.from_numpy(ws), torch.from_numpy(depth), torch.from_numpy(image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, ws, image)
        ctx.T_thresh = T_thresh
        return ws, depth, image

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        gs, gc = rm.composite_rays_train_backward(g_ws.contiguous().numpy(), g_image.contiguous().numpy(), sigmas.detach().numpy(), rgbs.detach().numpy(), deltas.numpy(), rays.numpy(), ws.numpy(), image.numpy(), ctx.T_thresh)
        return torch.from_numpy(gs), torch.from_numpy(gc), None, None, None


def run_cuda_train(rays_o, rays_d, message, P, S, dt_gamma=0.0, bg_color=1, max_steps=1024, T_thresh=1e-4, noises=None):
    """NeRFRenderer.run_cuda, training branch, renderer_wtmk.py:256-321.
    S: scene dict {bound, cascade, grid_size, density_bitfield (uint8 np), aabb (np[6]), min_near, density_scale}."""
    prefix = rays_o.shape[:-1]
    o, d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
    nears, fars = rm.near_far_from_aabb(o.numpy(), d.numpy(), S["aabb"], S["min_near"])
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = rm.march_rays_train(o.numpy(), d.numpy(), S["bound"], S["density_bitfield"], S["cascade"], S["grid_size"], nears, fars, counter, -1, noises is not None, 128, True, dt_gamma, max_steps, noises=noises)
    xyzs, dirs, deltas, rays = map(torch.from_numpy, (xyzs, dirs, deltas, rays))
    sigmas, rgbs = field_forward(xyzs, dirs, message, P)
    sigmas = S.get("density_scale", 1) * sigmas
    ws, depth, image = _CompositeTrain.apply(sigmas, rgbs, deltas, rays, T_thresh)
    image = image + (1 - ws).unsqueeze(-1) * bg_color
    nears_t, fars_t = torch.from_numpy(nears), torch.from_numpy(fars)
    depth = torch.clamp(depth - nears_t, min=0) / (fars_t - nears_t)
    return {"image": image.view(*prefix, 3), "depth": depth.view(*prefix), "weights_sum": ws,
            "rays": rays, "n_points": int(counter[0]), "xyzs": xyzs, "dirs": dirs, "deltas": deltas,
            "sigmas": sigmas, "rgbs": rgbs}


@torch.no_grad()
def run_cuda_eval(rays_o, rays_d, message, P, S, dt_gamma=0.0, bg_color=1, max_steps=1024, T_thresh=1e-4):
    """NeRFRenderer.run_cuda, inference branch, renderer_wtmk.py:323-377."""
    prefix = rays_o.shape[:-1]
    o, d = rays_o.contiguous().view(-1, 3).numpy(), rays_d.contiguous().view(-1, 3).numpy()
    N = o.shape[0]
    nears, fars = rm.near_far_from_aabb(o, d, S["aabb"], S["min_near"])
    ws, depth, image = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    step = 0
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = rm.march_rays(n_alive, n_step, rays_alive, rays_t, o, d, S["bound"], S["density_bitfield"], S["cascade"], S["grid_size"], nears, fars, 128, False, dt_gamma, max_steps)
        sigmas, rgbs = field_forward(torch.from_numpy(xyzs), torch.from_numpy(dirs), message, P)
        sigmas = S.get("density_scale", 1) * sigmas
        rm.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas.numpy(), rgbs.numpy(), deltas, ws, depth, image, T_thresh)
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])
        step += n_step
    ws_t, depth_t, image_t = torch.from_numpy(ws), torch.from_numpy(depth), torch.from_numpy(image)
    image_t = image_t + (1 - ws_t).unsqueeze(-1) * bg_color
    nears_t, fars_t = torch.from_numpy(nears), torch.from_numpy(fars)
    depth_t = torch.clamp(depth_t - nears_t, min=0) / (fars_t - nears_t)
    return {"image": image_t.view(*prefix, 3), "depth": depth_t.view(*prefix)}


def sample_pdf(bins, weights, n_samples, det=True, u=None):
    """renderer_wtmk.py:12-47 (NeRF's inverse-CDF sampling): bins [B,T], weights [B,T-1] -> [B,n_samples].  det=False takes the
    uniform draws `u` [B,n_samples] from the caller (the reference draws torch.rand)."""
    weights = weights + 1e-5                                    # :20
    pdf = weights / torch.sum(weights, -1, keepdim=True)        # :21
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)  # :23
    if det:                                                     # :25-27
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples).expand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)               # :33
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)        # :38-40 (the gather over an expanded copy, written directly)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)               # :43
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)                           # :44-45


def run_uniform(rays_o, rays_d, message, P, S, num_steps=512, bg_color=1, upsample_steps=0, training=True, aabb=None, u=None):
    """NeRFRenderer.run with perturb=False, renderer_wtmk.py:125-253.  upsample_steps > 0: the importance re-sampling branch :166-201 --
    note that the reference evaluates the re-sampled points with density(new_xyzs), i.e. message=None: the CLEAN field (:187)."""
    prefix = rays_o.shape[:-1]
    o, d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
    N = o.shape[0]
    box = np.asarray(S["aabb"] if aabb is None else aabb, np.float32)
    aabb_t = torch.from_numpy(box)
    nears, fars = rm.near_far_from_aabb(o.numpy(), d.numpy(), box, S["min_near"])
    nears, fars = torch.from_numpy(nears).unsqueeze(-1), torch.from_numpy(fars).unsqueeze(-1)
    z = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand(N, num_steps)
    z = nears + (fars - nears) * z
    sample_dist = (fars - nears) / num_steps
    clip = lambda p: torch.min(torch.max(p, aabb_t[:3]), aabb_t[3:])
    xyzs = clip(o.unsqueeze(-2) + d.unsqueeze(-2) * z.unsqueeze(-1))
    dn = density(xyzs.reshape(-1, 3), message, P)
    sigma, geo = dn["sigma"].view(N, num_steps), dn["geo_feat"].view(N, num_steps, -1)
    scale = S.get("density_scale", 1)
    T = num_steps
    if upsample_steps > 0:
        with torch.no_grad():
            deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)       # :169-170
            alphas = 1 - torch.exp(-deltas * scale * sigma.detach())
            shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
            weights = alphas * torch.cumprod(shifted, dim=-1)[..., :-1]
            z_mid = z[..., :-1] + 0.5 * deltas[..., :-1]                                                              # :177
            new_z = sample_pdf(z_mid, weights[:, 1:-1], upsample_steps, det=not training, u=u)                       # :178
            new_xyzs = clip(o.unsqueeze(-2) + d.unsqueeze(-2) * new_z.unsqueeze(-1))                                  # :180-181
        nd = density(new_xyzs.reshape(-1, 3), None, P)                                                                # :187: no message
        z, order = torch.sort(torch.cat([z, new_z], dim=1), dim=1)                                                    # :193-194
        xyzs = torch.gather(torch.cat([xyzs, new_xyzs], dim=1), 1, order.unsqueeze(-1).expand(-1, -1, 3))
        sigma = torch.gather(torch.cat([sigma, nd["sigma"].view(N, upsample_steps)], dim=1), 1, order)
        g_all = torch.cat([geo, nd["geo_feat"].view(N, upsample_steps, -1)], dim=1)
        geo = torch.gather(g_all, 1, order.unsqueeze(-1).expand(-1, -1, g_all.shape[-1]))
        T = num_steps + upsample_steps
    deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)
    alphas = 1 - torch.exp(-deltas * scale * sigma)
    shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    weights = alphas * torch.cumprod(shifted, dim=-1)[..., :-1]
    mask = (weights > 1e-4).reshape(-1)
    dirs = d.view(-1, 1, 3).expand_as(xyzs).reshape(-1, 3)
    rgbs = torch.zeros(N * T, 3)
    if mask.any():
        rgbs[mask] = color(dirs[mask], geo.reshape(N * T, -1)[mask], P)
    rgbs = rgbs.view(N, T, 3)
    ws = weights.sum(dim=-1)
    depth = torch.sum(weights * ((z - nears) / (fars - nears)).clamp(0, 1), dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2) + (1 - ws).unsqueeze(-1) * bg_color
    return {"image": image.view(*prefix, 3), "depth": depth.view(*prefix), "weights_sum": ws}


def render(rays_o, rays_d, message, P, S, staged=False, max_ray_batch=4096, cuda_ray=True, training=True, **kw):
    """NeRFRenderer.render, renderer_wtmk.py:541-575."""
    def _run(o, d):
        if not cuda_ray:
            return run_uniform(o, d, message, P, S, training=training, **{k: kw[k] for k in ("num_steps", "bg_color", "upsample_steps", "u") if k in kw})
        keys = ("dt_gamma", "bg_color", "max_steps", "T_thresh")
        fn = run_cuda_train if training else run_cuda_eval
        return fn(o, d, message, P, S, **{k: kw[k] for k in keys if k in kw})
    if not staged:
        return _run(rays_o, rays_d)
    B, N = rays_o.shape[:2]
    depth, image = torch.empty(B, N), torch.empty(B, N, 3)
    for b in range(B):
        for head in range(0, N, max_ray_batch):
            tail = min(head + max_ray_batch, N)
            r = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail])
            depth[b:b + 1, head:tail], image[b:b + 1, head:tail] = r["depth"].detach(), r["image"].detach()
    return {"depth": depth, "image": image}


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)  # hidden_models.py:13


def normalize_img(x):
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def distortion_layer(pred_rgb, distortion, draw=None):
    """Trainer.distortion_layer, nerf/utils_wtmk_disen.py:551-577, on the clamped blocks [B,H,W,3], with the call's random draw PASSED IN:
    noise      draw = the N(0, 0.1) tensor of :555 (`torch.normal(0, sqrt(0.1), size=pred_rgb.shape)`), added (:556);
    brightness draw = the factor f of torchvision ColorJitter(brightness=0.5) (:573-575; one f ~ U[0.5, 1.5] per call, `adjust_brightness` =
               blend with a black image, clamped to [0, 1] for float images);
    blurring   draw = the sigma of torchvision GaussianBlur(kernel_size=3, sigma=(0.01, 0.5)) (:568-570; one sigma ~ U[0.01, 0.5] per call):
               1-d kernel exp(-0.5 (x / sigma)^2) at x = linspace(-1, 1, 3), normalised; 2-d = outer product; reflect padding of 1; depthwise conv.
    rotation   draw = the angles in degrees, one per image (:558-560: torchvision RandomRotation((-30, 30)) called once per image; defaults
               InterpolationMode.NEAREST, expand=False, centre = image centre, fill 0): torchvision's tensor `rotate` builds the inverse affine
               matrix of -angle, [cos a, -sin a, 0; sin a, cos a, 0], applies it to the pixel-centre grid measured from the image centre and samples
               with grid_sample(mode='nearest', padding_mode='zeros', align_corners=False) -- restated with the same grid and the stock grid_sample,
               in float64 (away from rounding ties the float32 result is the same);
    scaling    draw = the factor sf (:563; one sf ~ U[0.75, 1.25] per call): F.interpolate(image [3, H, W], scale_factor=sf, mode='linear') per image
               (:565) -- the stock operator itself: a [3, H, W] tensor is a batch of 3 signals with H channels, resized along W to floor(W * sf).
    (torchvision is not installed in this image: its transforms are restated from their published implementation,
    torchvision/transforms/_functional_tensor.py `_blend` / `_get_gaussian_kernel1d` / `gaussian_blur` / `rotate` / `_gen_affine_grid`.)"""
    if distortion in (None, "none"):
        return pred_rgb                                                  # :553
    if distortion == "noise":
        return pred_rgb + draw                                           # :556
    x = pred_rgb.permute(0, 3, 1, 2)                                     # :567 / :572
    if distortion == "brightness":
        y = (float(draw) * x + (1.0 - float(draw)) * torch.zeros_like(x)).clamp(0, 1)
    elif distortion == "blurring":
        sigma = float(draw)
        k1 = torch.exp(-0.5 * (torch.linspace(-1.0, 1.0, 3, dtype=x.dtype) / sigma) ** 2)
        k1 = k1 / k1.sum()
        k2 = (k1[:, None] * k1[None, :]).expand(x.shape[1], 1, 3, 3)
        y = torch.nn.functional.conv2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect"), k2, groups=x.shape[1])
    elif distortion == "rotation":
        import math
        B, C, H, W = x.shape
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64) - (H - 1) / 2, torch.arange(W, dtype=torch.float64) - (W - 1) / 2, indexing="ij")
        grids = []
        for deg in draw:
            a = math.radians(float(deg))
            sx, sy = math.cos(a) * xs - math.sin(a) * ys, math.sin(a) * xs + math.cos(a) * ys          # source position, from the centre, in pixels
            grids.append(torch.stack([sx / (0.5 * W), sy / (0.5 * H)], dim=-1))                        # grid_sample's normalised coordinates (align_corners=False)
        y = torch.nn.functional.grid_sample(x.double(), torch.stack(grids), mode="nearest", padding_mode="zeros", align_corners=False).to(x.dtype)
    elif distortion == "scaling":
        y = torch.stack([torch.nn.functional.interpolate(image, scale_factor=float(draw), mode="linear") for image in x])      # :565
    else:
        raise NotImplementedError(distortion)
    return y.permute(0, 2, 3, 1)                                         # :571 / :576


def train_step(block_o, block_d, content_o, content_d, gt_rgb, message, P, S, decoder, lambda_w=1.0, lambda_i=1.0, distortion=None, draw=None, **kw):
    """Trainer.train_step with 3-channel images and loss_w='bce', nerf/utils_wtmk_disen.py:579-646 (+ :441 for the loss).
    distortion / draw: see distortion_layer."""
    out = run_cuda_train(block_o, block_d, message, P, S, bg_color=1, **kw)
    pred = torch.clamp(out["image"], min=0, max=1)                    # :592
    pred_dist = distortion_layer(pred, distortion, draw)               # :594
    decoded = decoder(normalize_img(pred_dist.permute(0, 3, 1, 2)))    # :595
    cont = run_cuda_train(content_o, content_d, message, P, S, bg_color=1, **kw)  # :616
    lossi = ((cont["image"] - gt_rgb) ** 2).mean()                     # :638 (MSELoss(reduction='none').mean())
    lossw = torch.nn.functional.binary_cross_entropy_with_logits(decoded * 10.0, message.unsqueeze(-1), reduction="mean")  # :441,641
    loss = lambda_w * lossw + lambda_i * lossi                         # :644
    return {"pred_rgb": pred, "content_pred_rgb": cont["image"], "decoded": decoded, "lossi": lossi, "lossw": lossw,
            "loss": loss, "block": out, "content": cont}


def bit_accuracy(pred, truth):
    """BIT_ACC.update, nerf/utils_wtmk_disen.py:340-343."""
    same = ~torch.logical_xor(pred > 0, truth > 0)
    return (torch.sum(same, dim=-1) / same.shape[-1])


def psnr(pred, truth):
    """PSNRMeter.update, nerf/utils_wtmk_disen.py:229-233."""
    p, t = np.asarray(pred), np.asarray(truth)
    return float(-10 * np.log10(np.mean((p - t) ** 2)))


def get_rays(poses, intrinsics, H, W, inds=None):
    """get_rays without the random index draw, nerf/utils_wtmk_disen.py:59-143 (inds given or all pixels)."""
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i = i.t().reshape(1, H * W).expand(B, H * W) + 0.5
    j = j.t().reshape(1, H * W).expand(B, H * W) + 0.5
    if inds is not None:
        i, j = torch.gather(i, -1, inds), torch.gather(j, -1, inds)
    zs = torch.ones_like(i)
    directions = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o, rays_d


# ------------------------------------------------------------------ density-grid maintenance

def _grid_cells(G, S):
    """Grid cells in S^3 blocks, x outermost (renderer_wtmk.py:393-408 / 463-472): yields (coords [n,3] int32, morton index [n] int64)."""
    from . import raymarch_ref as orm
    axis = torch.arange(G, dtype=torch.int32).split(S)
    for xs in axis:
        for ys in axis:
            for zs in axis:
                xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                yield coords, torch.from_numpy(orm.morton3D(coords.numpy())).long()


def mark_untrained_grid(state, poses, intrinsic, S=64):
    """renderer_wtmk.py:380-442.  state: {'density_grid' [C,G^3] torch fp32, 'bound', 'grid_size'}; cells seen by no camera get -1.
    Returns the number of marked cells."""
    grid, bound, G = state["density_grid"], state["bound"], state["grid_size"]
    C = grid.shape[0]
    poses = torch.as_tensor(poses)
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsic
    count = torch.zeros_like(grid)
    for coords, indices in _grid_cells(G, S):
        world = (2 * coords.float() / (G - 1) - 1).unsqueeze(0)                    # :411
        for cas in range(C):
            b = min(2 ** cas, bound)
            half = b / G
            cas_world = world * (b - half)                                         # :418
            head = 0
            while head < B:
                tail = min(head + S, B)
                cam = cas_world - poses[head:tail, :3, 3].unsqueeze(1)             # :426
                cam = cam @ poses[head:tail, :3, :3]
                mask_z = cam[:, :, 2] > 0
                mask_x = torch.abs(cam[:, :, 0]) < cx / fx * cam[:, :, 2] + half * 2
                mask_y = torch.abs(cam[:, :, 1]) < cy / fy * cam[:, :, 2] + half * 2
                count[cas, indices] += (mask_z & mask_x & mask_y).sum(0).reshape(-1)
                head += S
    grid[count == 0] = -1                                                          # :439
    return int((count == 0).sum())


def update_extra_state(state, density_fn, message=None, decay=0.95, S=128, rand_like=torch.rand_like, randint=torch.randint):
    """renderer_wtmk.py:445-538.  state: density_grid [C,G^3], density_bitfield, bound, grid_size, density_scale, density_thresh,
    iter_density, mean_density, step_counter [16,2] int32, local_step, mean_count.  density_fn(x [n,3], message) -> {'sigma': [n]}.
    `rand_like` / `randint`: the two draws the reference makes on its device generator (:478,488,492,514)."""
    from . import raymarch_ref as orm
    grid, bound, G = state["density_grid"], state["bound"], state["grid_size"]
    C = grid.shape[0]
    tmp = -torch.ones_like(grid)
    hits = torch.zeros(grid.shape, dtype=torch.int64)                              # (oracle only) how often a cell is written
    if state["iter_density"] < 16:                                                 # :456 full update
        for coords, indices in _grid_cells(G, S):
            xyzs = 2 * coords.float() / (G - 1) - 1
            for cas in range(C):
                b = min(2 ** cas, bound)
                half = b / G
                cas_xyzs = xyzs * (b - half)
                cas_xyzs += (rand_like(cas_xyzs) * 2 - 1) * half                   # :478
                sig = density_fn(cas_xyzs, message)["sigma"].reshape(-1).detach() * state["density_scale"]
                tmp[cas, indices] = sig
    else:                                                                          # :486 partial update
        N = G ** 3 // 4
        for cas in range(C):
            coords = randint(0, G, (N, 3))
            indices = torch.from_numpy(orm.morton3D(coords.int().numpy())).long()
            occ = torch.nonzero(grid[cas] > 0).squeeze(-1)
            occ = occ[randint(0, occ.shape[0], [N], dtype=torch.long)]
            occ_coords = torch.from_numpy(orm.morton3D_invert(occ.int().numpy()))
            indices = torch.cat([indices, occ], dim=0)
            coords = torch.cat([coords, occ_coords], dim=0)
            xyzs = 2 * coords.float() / (G - 1) - 1
            b = min(2 ** cas, bound)
            half = b / G
            cas_xyzs = xyzs * (b - half)
            cas_xyzs += (rand_like(cas_xyzs) * 2 - 1) * half
            sig = density_fn(cas_xyzs, message)["sigma"].reshape(-1).detach() * state["density_scale"]
            tmp[cas, indices] = sig                                                # :520 (a cell drawn twice keeps ONE of its values, which one is unspecified:
            hits[cas] += torch.bincount(indices, minlength=G ** 3)                 #  index_put_ with duplicates -- `hits` lets tests exclude those cells)
    valid = (grid >= 0) & (tmp >= 0)                                               # :522 EMA
    grid[valid] = torch.maximum(grid[valid] * decay, tmp[valid])
    state["mean_density"] = torch.mean(grid.clamp(min=0)).item()
    state["iter_density"] += 1
    thresh = min(state["mean_density"], state["density_thresh"])                   # :529
    state["density_bitfield"] = torch.from_numpy(orm.packbits(grid.numpy(), thresh))
    total = min(16, state["local_step"])                                           # :533
    if total > 0:
        state["mean_count"] = int(state["step_counter"][:total, 0].sum().item() / total)
    state["local_step"] = 0
    return tmp, hits
