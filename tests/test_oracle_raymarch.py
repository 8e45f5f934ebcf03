"""Property tests that protect oracle/raymarch_ref.c (the reference's CUDA file cannot run here,
so nothing else pins it).  CPU only."""
import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import raymarch_ref as rm


def test_morton_round_trip_and_bit_layout():
    rng = np.random.RandomState(0)
    c = rng.randint(0, 128, size=(5000, 3)).astype(np.int32)
    idx = rm.morton3D(c)
    assert idx.min() >= 0 and idx.max() < 128 ** 3
    np.testing.assert_array_equal(rm.morton3D_invert(idx), c)
    # x occupies bit 0, y bit 1, z bit 2 of every triple
    np.testing.assert_array_equal(rm.morton3D(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [2, 0, 0]], np.int32)), [1, 2, 4, 8])
    full = rm.morton3D(np.stack(np.meshgrid(*[np.arange(16)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.int32))
    assert len(np.unique(full)) == 16 ** 3 and full.max() == 16 ** 3 - 1


def test_packbits_matches_numpy_little_endian():
    rng = np.random.RandomState(1)
    grid = rng.randn(2, 4096).astype(np.float32)
    grid[0, :8] = 0.25  # equal to thresh -> not set (strict >)
    got = rm.packbits(grid, 0.25)
    want = np.packbits((grid.reshape(-1) > 0.25), bitorder="little")
    np.testing.assert_array_equal(got, want)


def test_near_far_slab_and_miss():
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    o = np.array([[0, 0, -3], [0, 0, -3], [0, 0, 0], [5, 5, 5]], np.float32)
    d = np.array([[0, 0, 1], [0, 1, 0], [0, 0, 1], [1e-3, 0, 1]], np.float32)
    with np.errstate(divide="ignore"):
        n, f = rm.near_far_from_aabb(o, d, aabb, 0.2)
    assert n[0] == 2.0 and f[0] == 4.0
    assert n[1] == f[1] == np.finfo(np.float32).max   # parallel ray outside the slab: miss
    assert n[2] == np.float32(0.2) and f[2] == 1.0      # origin inside: near clamped to min_near
    assert n[3] == f[3] == np.finfo(np.float32).max


def _scene_and_rays(n=256, bound=1.0, seed=0):
    grid, bitfield, C = cf.ball_scene(bound=bound)
    pose, intr, inds = cf.orbit_rays(n, seed=seed, radius=3.2248 if bound == 1.0 else 1.3)
    from oracle import field_ref as fr
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    return bitfield, C, o[0].contiguous().numpy(), d[0].contiguous().numpy()


@pytest.mark.parametrize("bound,dt_gamma", [(1.0, 0.0), (2.0, 0.0), (2.0, 1 / 128)])
def test_march_train_structure(bound, dt_gamma):
    bitfield, C, o, d = _scene_and_rays(bound=bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
    ctr = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, bound, bitfield, C, 128, nears, fars, ctr, -1, False, 128, True, dt_gamma, 1024)
    N = o.shape[0]
    assert ctr[1] == N and ctr[0] == rays[:, 2].sum() > 0
    np.testing.assert_array_equal(rays[:, 0], np.arange(N))
    np.testing.assert_array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:, 2])[:-1]]))
    m = int(ctr[0])
    assert xyzs.shape[0] == m + 128 - m % 128          # raymarching.py:225-226
    assert np.all(xyzs[m:] == 0) and np.all(deltas[m:] == 0)
    assert np.all(np.abs(xyzs) <= bound) and np.all(deltas[:m, 0] > 0) and np.all(deltas[:m, 1] >= deltas[:m, 0] * 0.999)
    # every sample of ray r carries ray r's direction and lies on the ray
    k = int(np.argmax(rays[:, 2]))
    off, cnt = rays[k, 1], rays[k, 2]
    np.testing.assert_array_equal(dirs[off:off + cnt], np.repeat(d[k:k + 1], cnt, 0))
    t = np.linalg.norm(xyzs[off:off + cnt] - o[k], axis=-1)
    assert np.all(np.diff(t) > 0)
    # count-only entry agrees
    rays2, total = rm.march_counts(o, d, bound, bitfield, C, 128, nears, fars, dt_gamma, 1024)
    np.testing.assert_array_equal(rays2, rays)
    assert total == m


def test_march_train_overflow_and_empty():
    bitfield, C, o, d = _scene_and_rays(n=64)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
    # mean_count bound smaller than needed: rays past the bound are dropped but still recorded (cu:416)
    xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, bitfield, C, 128, nears, fars, None, 200, False, 128, False, 0.0, 1024)
    assert xyzs.shape[0] == 256
    over = rays[:, 1] + rays[:, 2] > 256
    assert over.any()
    ws, depth, image = rm.composite_rays_train_forward(np.ones(256, np.float32), np.ones((256, 3), np.float32), deltas, rays)
    assert np.all(ws[rays[over, 0]] == 0)
    # an empty grid yields zero samples and zero outputs
    empty = np.zeros_like(bitfield)
    _, _, dl, r0 = rm.march_rays_train(o, d, 1.0, empty, C, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)
    assert r0[:, 2].sum() == 0 and dl.shape[0] == 128


def _cumprod_composite(sig, rgb, deltas, T_thresh):
    """renderer_wtmk.py:205-229 style formulation with the early-termination rule applied."""
    alpha = 1 - np.exp(-sig.astype(np.float64) * deltas[:, 0])
    T = np.concatenate([[1.0], np.cumprod(1 - alpha)])
    stop = np.nonzero(T[1:] < T_thresh)[0]
    n = len(sig) if len(stop) == 0 else stop[0] + 1
    w = (alpha * T[:-1])[:n]
    tt = np.cumsum(deltas[:n, 1].astype(np.float64))
    return w.sum(), (w * tt).sum(), (w[:, None] * rgb[:n]).sum(0)


@pytest.mark.parametrize("scale", [1.0, 400.0])
def test_composite_forward_matches_cumprod(scale):
    rng = np.random.RandomState(2)
    counts = np.array([0, 1, 7, 64, 200], np.int32)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    rays = np.stack([np.array([3, 0, 4, 1, 2], np.int32), offs, counts], -1)
    M = int(counts.sum())
    sig = (rng.rand(M) * scale).astype(np.float32)
    rgb = rng.rand(M, 3).astype(np.float32)
    deltas = np.stack([np.full(M, 0.0034, np.float32), rng.rand(M).astype(np.float32) * 0.01], -1)
    ws, depth, image = rm.composite_rays_train_forward(sig, rgb, deltas, rays, 1e-4)
    for rid, off, cnt in rays:
        w0, d0, i0 = _cumprod_composite(sig[off:off + cnt], rgb[off:off + cnt], deltas[off:off + cnt], 1e-4) if cnt else (0, 0, np.zeros(3))
        np.testing.assert_allclose(ws[rid], w0, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(depth[rid], d0, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(image[rid], i0, rtol=2e-5, atol=1e-7)


def test_composite_backward_matches_fp64_autograd():
    rng = np.random.RandomState(3)
    counts = np.array([5, 40], np.int32)
    rays = np.stack([np.array([1, 0], np.int32), np.array([0, 5], np.int32), counts], -1)
    M = 45
    sig = (rng.rand(M) * 30).astype(np.float32)
    rgb = rng.rand(M, 3).astype(np.float32)
    deltas = np.stack([np.full(M, 0.0034, np.float32), np.full(M, 0.0034, np.float32)], -1)
    ws, depth, image = rm.composite_rays_train_forward(sig, rgb, deltas, rays, 1e-4)
    g_ws = rng.randn(2).astype(np.float32)
    g_img = rng.randn(2, 3).astype(np.float32)
    gs, gc = rm.composite_rays_train_backward(g_ws, g_img, sig, rgb, deltas, rays, ws, image, 1e-4)
    s64 = torch.tensor(sig, dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(rgb, dtype=torch.float64, requires_grad=True)
    loss = 0
    for rid, off, cnt in rays:
        a = 1 - torch.exp(-s64[off:off + cnt] * 0.0034)
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - a]), 0)[:-1]
        w = a * T
        loss = loss + g_ws[rid] * w.sum() + (torch.tensor(g_img[rid], dtype=torch.float64) * (w[:, None] * c64[off:off + cnt]).sum(0)).sum()
    loss.backward()
    np.testing.assert_allclose(gs, s64.grad.numpy(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(gc, c64.grad.numpy(), rtol=2e-4, atol=1e-6)


def test_eval_march_composite_agree_with_train_path_when_thin():
    """With a thin medium (no early termination) the burst-wise eval kernels integrate the same samples."""
    bitfield, C, o, d = _scene_and_rays(n=32)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, bitfield, C, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)
    sig_of = lambda p: 0.5 + 0.25 * p[:, 0]
    rgb_of = lambda p: np.stack([0.5 + 0.4 * p[:, 1], 0.3 + 0 * p[:, 1], 0.5 - 0.4 * p[:, 2]], -1).astype(np.float32)
    ws, depth, image = rm.composite_rays_train_forward(sig_of(xyzs), rgb_of(xyzs), deltas, rays, 1e-4)
    N = 32
    ws2, depth2, image2 = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive, rays_t, step = np.arange(N, dtype=np.int32), nears.copy(), 0
    while step < 1024 and alive.shape[0] > 0:
        n_alive = alive.shape[0]
        n_step = max(min(N // n_alive, 8), 1)
        p, dd, dl = rm.march_rays(n_alive, n_step, alive, rays_t, o, d, 1.0, bitfield, C, 128, nears, fars, 128, False, 0.0, 1024)
        rm.composite_rays(n_alive, n_step, alive, rays_t, sig_of(p), rgb_of(p), dl, ws2, depth2, image2, 1e-4)
        alive = np.ascontiguousarray(alive[alive >= 0])
        step += n_step
    np.testing.assert_allclose(ws2, ws, rtol=0, atol=2e-6)
    np.testing.assert_allclose(image2, image, rtol=0, atol=2e-6)
