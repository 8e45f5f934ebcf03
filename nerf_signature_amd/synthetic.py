"""Seeded synthetic inputs of the shapes BASELINE.json names: there is no dataset or
checkpoint in the build/benchmark environment, so scenes, cameras, rays, tables and MLP weights are
generated here.  Pure numpy/torch on the host; nothing in this file touches the oracle."""
import math

import numpy as np
import torch

SCENES = {
    # name: bound, camera radius, image H, W, focal, dt_gamma, message_dim, block grid (rows, cols)
    "hotdog": dict(bound=1.0, radius=3.2248, H=400, W=400, focal=555.56, dt_gamma=0.0, message_dim=32, rows=32, cols=32, inside=False),
    "counter": dict(bound=2.0, radius=1.3, H=400, W=400, focal=555.56 * 0.5, dt_gamma=0.0, message_dim=32, rows=32, cols=32, inside=True),
    "fern": dict(bound=2.0, radius=1.6, H=756, W=1008, focal=815.0, dt_gamma=1.0 / 128, message_dim=48, rows=64, cols=64, inside=True),
}


def morton3d_np(coords):
    """10-bit-per-axis interleave, x -> bit 0 (numpy counterpart of rm_morton3D for host-side scene setup)."""
    def spread(v):
        v = v.astype(np.uint32)
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    return (spread(coords[:, 0]) | (spread(coords[:, 1]) << np.uint32(1)) | (spread(coords[:, 2]) << np.uint32(2))).astype(np.int64)


def density_grid(bound, H=128, ball_radius=0.5, shell=(1.5, 2.0)):
    """[C, H^3] fp32 in morton order: density 100 inside a centred ball on cascade 0; for bound > 1 the outer
    cascade additionally holds a shell 1.5 < |p|_inf < 2 (scene S1/S2: exercises the coarse cascade)."""
    C = 1 + math.ceil(math.log2(bound))
    ii = np.arange(H, dtype=np.int64)
    coords = np.stack(np.meshgrid(ii, ii, ii, indexing="ij"), -1).reshape(-1, 3)
    idx = morton3d_np(coords)
    grid = np.zeros((C, H ** 3), np.float32)
    for c in range(C):
        extent = min(2 ** c, bound)
        p = (2 * (coords.astype(np.float32) + 0.5) / H - 1) * extent
        occ = np.linalg.norm(p, axis=-1) < ball_radius
        if c > 0:
            linf = np.abs(p).max(-1)
            occ |= (linf > shell[0]) & (linf < shell[1])
        grid[c, idx] = np.where(occ, 100.0, 0.0)
    return grid


def pack_bits_np(grid, density_thresh=10.0):
    thresh = min(float(np.clip(grid, 0, None).mean()), density_thresh)
    return np.packbits((grid.reshape(-1) > thresh), bitorder="little"), thresh


def orbit_pose(theta, phi, radius):
    """Camera-to-world pose on a sphere, looking at the origin (OpenCV-style axes: x right, y down, z forward)."""
    c = np.array([radius * math.sin(theta) * math.sin(phi), radius * math.cos(theta), radius * math.sin(theta) * math.cos(phi)], np.float32)
    fwd = -c / np.linalg.norm(c)
    right = np.cross(fwd, np.array([0, 1, 0], np.float32))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, -up, fwd, c
    return pose


def get_rays(poses, intrinsics, H, W, inds=None):
    """Pinhole rays through pixel centres (the arithmetic of get_rays, nerf/utils_wtmk_disen.py:59-143, for given
    pixel indices).  poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy).  Returns rays_o, rays_d [B,N,3]."""
    B, dev = poses.shape[0], poses.device
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=dev), torch.linspace(0, H - 1, H, device=dev), indexing="ij")
    i = i.t().reshape(1, H * W).expand(B, H * W) + 0.5
    j = j.t().reshape(1, H * W).expand(B, H * W) + 0.5
    if inds is not None:
        i, j = torch.gather(i, -1, inds), torch.gather(j, -1, inds)
    zs = torch.ones_like(i)
    dirs = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    rays_d = dirs @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o.contiguous(), rays_d.contiguous()


def content_rays(scene, n_rays, seed, device="cpu"):
    """One orbit camera, n_rays seeded random pixels (`inds = randint(0, H*W)`, nerf/utils_wtmk_disen.py:105)."""
    cfg = SCENES[scene]
    rng = np.random.RandomState(seed)
    pose = orbit_pose(0.6 + 0.9 * rng.rand(), 2 * math.pi * rng.rand(), cfg["radius"])
    intr = (cfg["focal"], cfg["focal"], cfg["W"] / 2, cfg["H"] / 2)
    inds = torch.from_numpy(rng.randint(0, cfg["H"] * cfg["W"], size=n_rays).astype(np.int64))[None]
    o, d = get_rays(torch.from_numpy(pose)[None], intr, cfg["H"], cfg["W"], inds)
    return o.to(device), d.to(device)


def block_rays(scene, device="cpu"):
    """Rays of message_dim image blocks of the watermark pose, [D, bh, bw, 3] (nerf/provider_wtmk.py:470-494).

    The reference keeps the D blocks whose rendering JPEG-compresses worst, i.e. the most textured ones
    (provider_wtmk.py:146-218); the deterministic stand-in keeps the D blocks with the largest fraction of
    rays hitting the occupied ball, nearest the silhouette centre first."""
    cfg = SCENES[scene]
    H, W, D = cfg["H"], cfg["W"], cfg["message_dim"]
    bh, bw = H // cfg["rows"], W // cfg["cols"]
    pose = orbit_pose(1.1, 0.7, cfg["radius"])
    intr = (cfg["focal"], cfg["focal"], W / 2, H / 2)
    o, d = get_rays(torch.from_numpy(pose)[None], intr, H, W)
    o, d = o.view(H, W, 3), d.view(H, W, 3)
    # ray / ball(0.5) intersection test
    b = (o * d).sum(-1)
    hit = (b * b - ((o * o).sum(-1) - 0.25)) > 0
    hit = hit[:bh * cfg["rows"], :bw * cfg["cols"]].reshape(cfg["rows"], bh, cfg["cols"], bw).float().mean(dim=(1, 3))
    rr, cc = torch.meshgrid(torch.arange(cfg["rows"]), torch.arange(cfg["cols"]), indexing="ij")
    dist = ((rr + 0.5) * bh - H / 2) ** 2 + ((cc + 0.5) * bw - W / 2) ** 2
    order = sorted(range(cfg["rows"] * cfg["cols"]), key=lambda k: (-float(hit.view(-1)[k]), float(dist.view(-1)[k]), k))[:D]
    bo = torch.stack([o[(k // cfg["cols"]) * bh:(k // cfg["cols"] + 1) * bh, (k % cfg["cols"]) * bw:(k % cfg["cols"] + 1) * bw] for k in order])
    bd = torch.stack([d[(k // cfg["cols"]) * bh:(k // cfg["cols"] + 1) * bh, (k % cfg["cols"]) * bw:(k % cfg["cols"] + 1) * bw] for k in order])
    return bo.contiguous().to(device), bd.contiguous().to(device)


def table_values(index, scale, T=1 << 19):
    """[T,2] fp32 table, a closed-form integer hash of (row, feature, table index) mapped to [-scale, scale)."""
    v = (np.arange(T * 2, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(index) * np.uint64(0x9E3779B9) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    v = (v ^ (v >> np.uint64(15))) * np.uint64(2246822519) & np.uint64(0xFFFFFFFF)
    v = v ^ (v >> np.uint64(13))
    return ((v.astype(np.float64) / 4294967296.0 - 0.5) * 2 * scale).astype(np.float32).reshape(T, 2)


@torch.no_grad()
def init_model(model, scene, codebook_scale=0.05, opaque=False):
    """Random-init weights of the reference's architecture + the scene's occupancy grid, written into `model`
    (a NeRFNetwork): base tables U(-0.5,0.5), codebook U(-s,s) ("trained-like"; the reference initialises at
    1e-4, hash_encoding_wtmk_bit.py:69), MLPs as constructed (Xavier, seeds 1337/1338).  opaque=True scales the
    density head so that transmittance drops below 1e-4 inside the ball (exercises early termination)."""
    cfg = SCENES[scene]
    for l, emb in enumerate(model.encoder.embeddings):
        emb.weight.copy_(torch.from_numpy(table_values(l, 0.5)))
    for l, emb in enumerate(model.msg_encoder.embeddings):
        emb.weight.copy_(torch.from_numpy(table_values(100 + l, codebook_scale)))
    if opaque:
        model.sigma_net.params[2048:2048 + 64] *= 8.0
    grid = density_grid(cfg["bound"])
    bits, thresh = pack_bits_np(grid, model.density_thresh)
    model.density_grid.copy_(torch.from_numpy(grid))
    model.density_bitfield.copy_(torch.from_numpy(bits))
    model.mean_density = float(np.clip(grid, 0, None).mean())
    return model
