"""GPU parity of the ray-marching kernels against the CPU oracle, through the C ABI (ctypes -> libnerfsig.so)."""
import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr
from oracle import raymarch_ref as orm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rmod():
    from nerf_signature_amd import raymarching
    return raymarching


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(n, bound, seed=0, radius=None):
    grid, bitfield, C = cf.ball_scene(bound=bound)
    pose, intr, inds = cf.orbit_rays(n, seed=seed, radius=radius or (3.2248 if bound == 1.0 else 1.3))
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    return grid, bitfield, C, o[0].contiguous().numpy(), d[0].contiguous().numpy()


def test_utils_bit_exact(rmod):
    rng = np.random.RandomState(0)
    c = rng.randint(0, 128, size=(10007, 3)).astype(np.int32)
    idx = rmod.morton3D(_cuda(c))
    np.testing.assert_array_equal(idx.cpu().numpy(), orm.morton3D(c))
    np.testing.assert_array_equal(rmod.morton3D_invert(idx).cpu().numpy(), c)
    grid = rng.randn(2, 128 ** 3 // 16).astype(np.float32)
    np.testing.assert_array_equal(rmod.packbits(_cuda(grid), 0.1).cpu().numpy(), orm.packbits(grid, 0.1))
    odd = rng.randn(1, 8 * 37).astype(np.float32)  # byte count not a multiple of 4
    np.testing.assert_array_equal(rmod.packbits(_cuda(odd), 0.0).cpu().numpy(), orm.packbits(odd, 0.0))


@pytest.mark.parametrize("bound", [1.0, 2.0])
def test_near_far_bit_exact(rmod, bound):
    _, _, _, o, d = _scene(4096, bound)
    d[5] = (0, 1, 0)     # axis-parallel ray (infinite reciprocal)
    o[6] = (9, 9, 9)     # misses the box
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        n0, f0 = orm.near_far_from_aabb(o, d, aabb, 0.2)
    n1, f1 = rmod.near_far_from_aabb(_cuda(o), _cuda(d), _cuda(aabb), 0.2)
    np.testing.assert_array_equal(n1.cpu().numpy(), n0)
    np.testing.assert_array_equal(f1.cpu().numpy(), f0)


def test_sph_from_ray(rmod):
    _, _, _, o, d = _scene(1024, 1.0)
    want = orm.sph_from_ray(o * 0.1, d, 4.0)
    got = rmod.sph_from_ray(_cuda(o * 0.1), _cuda(d), 4.0)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=2e-6)


@pytest.mark.parametrize("bound,dt_gamma,n", [(1.0, 0.0, 4096), (2.0, 0.0, 2048), (2.0, 1 / 128, 2048), (1.0, 0.0, 77), (8.0, 1 / 128, 1024), (16.0, 1 / 256, 1024),
                                              (1.5, 0.0, 1024)])
def test_march_train_counts_and_points_bit_exact(rmod, bound, dt_gamma, n):
    _, bitfield, C, o, d = _scene(n, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = orm.near_far_from_aabb(o, d, aabb, 0.2)
    ctr0 = np.zeros(2, np.int32)
    x0, d0, dl0, rays0 = orm.march_rays_train(o, d, bound, bitfield, C, 128, nears, fars, ctr0, -1, False, 128, True, dt_gamma, 1024)
    ctr1 = torch.zeros(2, dtype=torch.int32, device="cuda")
    x1, d1, dl1, rays1 = rmod.march_rays_train(_cuda(o), _cuda(d), bound, _cuda(bitfield), C, 128, _cuda(nears), _cuda(fars), ctr1,
                                               -1, False, 128, True, dt_gamma, 1024)
    np.testing.assert_array_equal(ctr1.cpu().numpy(), ctr0)           # total point count, ray count
    np.testing.assert_array_equal(rays1.cpu().numpy(), rays0)         # (id, offset, count) per ray: bit-exact
    assert ctr0[0] > 0
    np.testing.assert_array_equal(x1.cpu().numpy(), x0)
    np.testing.assert_array_equal(d1.cpu().numpy(), d0)
    np.testing.assert_array_equal(dl1.cpu().numpy(), dl0)


def test_march_train_perturbed_and_bounded(rmod):
    """Fixed noise vector through the C ABI directly (the wrapper draws torch.rand) and the mean_count-bounded mode."""
    from nerf_signature_amd import _native as nv
    _, bitfield, C, o, d = _scene(512, 1.0, seed=3)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = orm.near_far_from_aabb(o, d, aabb, 0.2)
    noises = np.random.RandomState(1).rand(512).astype(np.float32)
    ctr0 = np.zeros(2, np.int32)
    x0, d0, dl0, rays0 = orm.march_rays_train(o, d, 1.0, bitfield, C, 128, nears, fars, ctr0, 3000, True, 128, False, 0.0, 1024, noises=noises)
    M = x0.shape[0]
    assert M == 3072 and ctr0[0] > M      # some rays overflow the bound and are dropped
    ctr1 = torch.zeros(2, dtype=torch.int32, device="cuda")
    to = lambda a: _cuda(a)
    _, _, rays1, write = rmod.march_rays_train_device(to(o), to(d), 1.0, to(bitfield), C, 128, to(nears), to(fars), ctr1, to(noises), 0.0, 1024)
    x1, d1, dl1 = write(M)
    np.testing.assert_array_equal(rays1.cpu().numpy(), rays0)
    np.testing.assert_array_equal(x1.cpu().numpy(), x0)
    np.testing.assert_array_equal(dl1.cpu().numpy(), dl0)


def test_composite_train_fwd_bwd(rmod):
    _, bitfield, C, o, d = _scene(2048, 1.0)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = orm.near_far_from_aabb(o, d, aabb, 0.2)
    x0, d0, dl0, rays0 = orm.march_rays_train(o, d, 1.0, bitfield, C, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)
    rng = np.random.RandomState(4)
    M = x0.shape[0]
    for scale in (2.0, 300.0):   # thin medium, and opaque enough to hit the T < 1e-4 early exit
        sig = (rng.rand(M) * scale).astype(np.float32)
        rgb = rng.rand(M, 3).astype(np.float32)
        ws0, dep0, img0 = orm.composite_rays_train_forward(sig, rgb, dl0, rays0, 1e-4)
        s1, c1 = _cuda(sig).requires_grad_(True), _cuda(rgb).requires_grad_(True)
        ws1, dep1, img1 = rmod.composite_rays_train(s1, c1, _cuda(dl0), _cuda(rays0), 1e-4)
        np.testing.assert_allclose(ws1.detach().cpu().numpy(), ws0, rtol=0, atol=2e-6)
        np.testing.assert_allclose(dep1.detach().cpu().numpy(), dep0, rtol=0, atol=1e-5)
        np.testing.assert_allclose(img1.detach().cpu().numpy(), img0, rtol=0, atol=2e-6)
        g_ws, g_img = rng.randn(2048).astype(np.float32), rng.randn(2048, 3).astype(np.float32)
        gs0, gc0 = orm.composite_rays_train_backward(g_ws, g_img, sig, rgb, dl0, rays0, ws0, img0, 1e-4)
        torch.autograd.backward([ws1, img1], [_cuda(g_ws), _cuda(g_img)])
        np.testing.assert_allclose(c1.grad.cpu().numpy(), gc0, rtol=0, atol=2e-6)
        np.testing.assert_allclose(s1.grad.cpu().numpy(), gs0, rtol=1e-4, atol=2e-6 * max(1.0, 3.0 / scale))


def test_eval_march_composite_compact(rmod):
    _, bitfield, C, o, d = _scene(1000, 1.0)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = orm.near_far_from_aabb(o, d, aabb, 0.2)
    N = 1000
    sig_of = lambda p: (40.0 * (0.5 + p[:, 0])).astype(np.float32)
    rgb_of = lambda p: np.stack([0.5 + 0.4 * p[:, 1], 0.3 + 0 * p[:, 1], 0.5 - 0.4 * p[:, 2]], -1).astype(np.float32)
    ws0, dep0, img0 = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive0, t0 = np.arange(N, dtype=np.int32), nears.copy()
    ws1, dep1, img1 = (torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, 3, device="cuda"))
    alive1, t1 = torch.arange(N, dtype=torch.int32, device="cuda"), _cuda(nears).clone()
    bf, oc, dc, nc, fc = _cuda(bitfield), _cuda(o), _cuda(d), _cuda(nears), _cuda(fars)
    step = 0
    while step < 1024 and alive0.shape[0] > 0:
        n_alive = alive0.shape[0]
        n_step = max(min(N // n_alive, 8), 1)
        p0, dd0, dl0 = orm.march_rays(n_alive, n_step, alive0, t0, o, d, 1.0, bitfield, C, 128, nears, fars, 128, False, 0.0, 1024)
        p1, dd1, dl1 = rmod.march_rays(n_alive, n_step, alive1, t1, oc, dc, 1.0, bf, C, 128, nc, fc, 128, False, 0.0, 1024)
        np.testing.assert_array_equal(p1.cpu().numpy(), p0)
        np.testing.assert_array_equal(dl1.cpu().numpy(), dl0)
        orm.composite_rays(n_alive, n_step, alive0, t0, sig_of(p0), rgb_of(p0), dl0, ws0, dep0, img0, 1e-2)
        rmod.composite_rays(n_alive, n_step, alive1, t1, _cuda(sig_of(p0)), _cuda(rgb_of(p0)), dl1, ws1, dep1, img1, 1e-2)
        np.testing.assert_array_equal(alive1.cpu().numpy(), alive0)      # which rays terminated: bit-exact
        alive0 = np.ascontiguousarray(alive0[alive0 >= 0])
        out, n_out = rmod.compact_alive(alive1)
        assert int(n_out.item()) == alive0.shape[0]
        alive1 = out[:alive0.shape[0]].contiguous()
        np.testing.assert_array_equal(alive1.cpu().numpy(), alive0)
        step += n_step
    np.testing.assert_allclose(ws1.cpu().numpy(), ws0, rtol=0, atol=5e-6)
    np.testing.assert_allclose(img1.cpu().numpy(), img0, rtol=0, atol=5e-6)
    np.testing.assert_allclose(dep1.cpu().numpy(), dep0, rtol=0, atol=2e-5)


def test_argument_checks_raise(rmod):
    from nerf_signature_amd import _native as nv
    with pytest.raises(ValueError):
        rmod.march_rays_train(torch.zeros(4, 3, device="cuda"), torch.ones(4, 3, device="cuda"), 1.0, torch.zeros(10, dtype=torch.uint8, device="cuda"),
                              1, 128, torch.zeros(4, device="cuda"), torch.ones(4, device="cuda"), None, -1, False, 128, True, 0.0, 1024)
    with pytest.raises(ValueError):
        nv.call("rm_morton3D", None, 4, None, None)


def test_get_rays_on_device_matches_reference_golden():
    """rg_get_rays vs the reference's own get_rays output (golden G7) and vs the oracle for whole images / batches."""
    import os
    from nerf_signature_amd import rays
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_get_rays.npz"))
    poses = torch.from_numpy(g["pose"])[None].cuda()
    torch.manual_seed(0)
    out = rays.get_rays(poses, g["intrinsics"], 400, 400, N=-1)
    assert out["rays_o"].shape == (1, 160000, 3) and torch.equal(out["inds"][0].cpu(), torch.arange(160000))
    np.testing.assert_array_equal(out["rays_o"][0, g["inds"]].cpu().numpy(), g["rays_o"])
    np.testing.assert_allclose(out["rays_d"][0, g["inds"]].cpu().numpy(), g["rays_d"], rtol=0, atol=5e-7)   # 3-term dot product: order/FMA differ by an ulp or two
    sub = rays.get_rays(poses, g["intrinsics"], 400, 400, N=4096)
    assert sub["rays_d"].shape == (1, 4096, 3) and sub["inds"].shape == (1, 4096) and int(sub["inds"].max()) < 160000
    np.testing.assert_array_equal(sub["rays_d"][0].cpu().numpy(), out["rays_d"][0, sub["inds"][0]].cpu().numpy())
    two = torch.cat([poses, poses.roll(1, 2)], 0)
    o0, d0 = fr.get_rays(two.cpu(), g["intrinsics"], 30, 50)
    o1 = rays.get_rays(two, g["intrinsics"], 30, 50)
    np.testing.assert_allclose(o1["rays_d"].cpu().numpy(), d0.numpy(), rtol=0, atol=5e-7)
    np.testing.assert_array_equal(o1["rays_o"].cpu().numpy(), o0.numpy())
    patch = rays.get_rays(poses, g["intrinsics"], 400, 400, N=4096, patch_size=16)
    assert patch["rays_d"].shape == (1, 4096, 3)
    em = torch.rand(1, 128 * 128).cuda()
    coarse = rays.get_rays(poses, g["intrinsics"], 400, 400, N=1024, error_map=em)
    assert coarse["inds_coarse"].shape == (1, 1024) and coarse["rays_d"].shape == (1, 1024, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("capacity_frac", [None, 0.6])
def test_composite_with_fused_tail_matches_separate_launches(capacity_frac):
    """rm_composite_train_finish_fwd/_bwd == composite_rays_train followed by the render tail, values and gradients -- including the
    rows the fused backward zero-fills itself: samples after early termination, rays dropped by a too-small point buffer, padding."""
    import torch
    from nerf_signature_amd import raymarching as rm
    from nerf_signature_amd.renderer import _CompositeFinish, _Finish
    dev = torch.device("cuda")
    _, bitfield, _, o, d = _scene(700, 1.0, seed=2, radius=1.6)
    bits, o, d = _cuda(bitfield), _cuda(o), _cuda(d)
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)
    nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    if capacity_frac is None:
        xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, counter, -1, False, 128, True, 0.0, 1024)
    else:
        full = rm.march_rays_train(o, d, 1.0, bits, 1, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)[0].shape[0]
        xyzs, dirs, deltas, rays = rm.march_rays_train_capacity(o, d, 1.0, bits, 1, 128, nears, fars, counter, rm.padded_point_count(int(full * capacity_frac)))
        assert int(counter[0]) > xyzs.shape[0]                     # some rays did not fit
    M = xyzs.shape[0]
    torch.manual_seed(0)
    sig = (torch.rand(M, device=dev) * 60).requires_grad_(True)     # dense enough for early termination (T < 1e-4) inside rays
    rgb = torch.rand(M, 3, device=dev).requires_grad_(True)
    bg = torch.rand(o.shape[0], 3, device=dev)
    gi, gw = torch.randn(o.shape[0], 3, device=dev), torch.randn(o.shape[0], device=dev)
    ws0, dp0, im0 = rm.composite_rays_train(sig, rgb, deltas, rays, 1e-4)
    im0f, dp0f = _Finish.apply(im0, dp0, ws0, nears, fars, bg)
    g0 = torch.autograd.grad([im0f, ws0], [sig, rgb], [gi, gw])
    junk = torch.full((4 * M + 1024,), float("nan"), device=dev)   # so that recycled allocations do not happen to hold zeros
    del junk
    ws1, dp1, im1 = _CompositeFinish.apply(sig, rgb, deltas, rays, nears, fars, bg, 1e-4)
    g1 = torch.autograd.grad([im1, ws1], [sig, rgb], [gi, gw])
    assert torch.equal(ws0, ws1) and torch.equal(im0f, im1)
    assert torch.equal(torch.nan_to_num(dp0f), torch.nan_to_num(dp1))
    for a, b in zip(g0, g1):
        assert torch.isfinite(b).all()
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-6, atol=1e-7)
    terminated = (g0[1].abs().sum(dim=1) == 0) & (deltas[:, 0] > 0)
    assert bool(terminated.any())                                   # the early-termination rows were exercised


def test_device_ray_sampler_matches_get_rays():
    """rg_sample_rays: pose (step * stride + offset) mod P from a device counter, uniformly drawn pixels, rays identical to rg_get_rays for
    the same (pose, indices), ground truth = the stored image's pixels; a new step draws new pixels, the same step the same ones."""
    from nerf_signature_amd import rays
    P, H, W, N = 5, 60, 80, 4096
    intr = (70.0, 72.0, W / 2, H / 2)
    poses = torch.stack([torch.from_numpy(cf.orbit_rays(1, seed=0, radius=2.0 + 0.1 * k)[0]) for k in range(P)]).cuda()
    images = torch.rand(P, H * W, 3, device="cuda")
    s = rays.DeviceRaySampler(poses, images, intr, H, W, N, stride=2, offset=1, seed=99)
    ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
    o, d, gt = (torch.empty(1, N, 3, device="cuda") for _ in range(3))
    inds, pose = torch.empty(N, dtype=torch.int64, device="cuda"), torch.empty(1, dtype=torch.int32, device="cuda")
    seen = []
    for step in (0, 1, 2, 7, 7):
        ctr.fill_(step)
        s.sample_into(ctr, o, d, gt, inds, pose)
        k = (step * 2 + 1) % P
        assert int(pose) == k and int(inds.min()) >= 0 and int(inds.max()) < H * W
        want = rays.get_rays(poses[k:k + 1], intr, H, W, N=-1)
        assert torch.equal(o[0], want["rays_o"][0, inds]) and torch.equal(d[0], want["rays_d"][0, inds])
        assert torch.equal(gt[0], images[k][inds])
        seen.append(inds.clone())
    assert torch.equal(seen[3], seen[4]) and not torch.equal(seen[0], seen[1])
    # uniform: every quarter of the image receives its share (4096 draws: +-5 sigma)
    q = torch.bincount((seen[0] * 4 // (H * W)), minlength=4).float()
    assert float((q - N / 4).abs().max()) < 5 * (N * 0.25 * 0.75) ** 0.5
    with pytest.raises(ValueError):
        s.sample_into(ctr, o[:, :10], d, gt)


@pytest.mark.parametrize("bound,dt_gamma,n,perturb", [(1.0, 0.0, 4608, False), (1.0, 0.0, 1, False), (1.0, 0.0, 63, True), (2.0, 0.0, 8704, False),
                                                      (2.0, 1.0 / 128, 4096, True), (1.0, 0.0, 12288, False)])
def test_two_enqueue_march_equals_the_four_enqueue_march(rmod, bound, dt_gamma, n, perturb):
    """The captured step's march -- rm_march_train_count_nf (the walk computes near / far itself) + rm_march_train_scan_write (every
    workgroup scans the counts for itself, offsets in LDS) -- against near_far + count + scan + write: limits, ray table, totals, every
    row and the zero padding bit for bit; a capacity smaller than the total drops the same rays (raymarching.cu:416); and against the C
    oracle's counts."""
    from nerf_signature_amd import _native as nv
    _, bitfield, C, o, d = _scene(n, bound, seed=3)
    oc, dc, bits = _cuda(o), _cuda(d), _cuda(bitfield)
    aabb = torch.tensor([-bound] * 3 + [bound] * 3, device="cuda")
    noises = torch.rand(n, device="cuda") if perturb else None
    nears, fars = rmod.near_far_from_aabb(oc, dc, aabb, 0.2)
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
    counts, t_rec, rays, write = rmod.march_rays_train_device(oc, dc, bound, bits, C, 128, nears, fars, ctr, noises, dt_gamma, 1024)
    total = int(ctr[0])
    rays0, total0 = orm.march_counts(o, d, bound, bitfield, C, 128, nears.cpu().numpy(), fars.cpu().numpy(), dt_gamma, 1024,
                                     noises=None if noises is None else noises.cpu().numpy())
    assert total == total0 and np.array_equal(rays[:, 2].cpu().numpy(), rays0[:, 2])
    assert n <= rmod.scan_write_max_rays() == 12288
    for M in (rmod.padded_point_count(total), max(128, (total // 2) // 128 * 128), 0):
        x4, d4, dl4 = write(M) if M else (None, None, None)
        nears2, fars2 = torch.full((n,), -1.0, device="cuda"), torch.full((n,), -1.0, device="cuda")
        ctr2 = torch.full((2,), -5, dtype=torch.int32, device="cuda")
        out = None
        if M:
            out = (torch.full((M, 3), 7.0, device="cuda"), torch.full((M, 3), 7.0, device="cuda"), torch.full((M, 2), 7.0, device="cuda"),
                   torch.full((n, 3), -9, dtype=torch.int32, device="cuda"))
        _, _, rays2, write2 = rmod.march_rays_train_device(oc, dc, bound, bits, C, 128, nears2, fars2, ctr2, noises, dt_gamma, 1024, capacity=M, out=out,
                                                           limits=(aabb, 0.2))
        if M:
            x2, d2, dl2 = write2(M)
            assert torch.equal(x2, x4) and torch.equal(d2, d4) and torch.equal(dl2, dl4)
        else:      # no rows: the launch still leaves the ray table and the totals
            empty = torch.empty(0, 3, device="cuda")
            nv.call("rm_march_train_scan_write", nv.ptr(oc), nv.ptr(dc), float(bound), float(dt_gamma), 1024, n, C, 128, 0, nv.ptr(nears2), nv.ptr(noises),
                    nv.ptr(t_rec), nv.ptr(counts), nv.ptr(rays2), nv.ptr(ctr2), None, None, None, nv.stream())
            del empty
        assert torch.equal(nears2, nears) and torch.equal(fars2, fars)
        assert torch.equal(rays2, rays) and torch.equal(ctr2, ctr)
    with pytest.raises(ValueError, match="outside"):
        nv.call("rm_march_train_scan_write", nv.ptr(oc), nv.ptr(dc), float(bound), float(dt_gamma), 1024, 12289, C, 128, 128, nv.ptr(nears), None,
                nv.ptr(t_rec), nv.ptr(counts), nv.ptr(rays), nv.ptr(ctr), nv.ptr(oc), nv.ptr(oc), nv.ptr(oc), nv.stream())


@pytest.mark.parametrize("n", [1, 4095, 4096, 4097, 20000, 262144])
def test_wide_scan_equals_the_single_workgroup_scan(n):
    """rm_march_train_scan_wide (two launches of 4096-ray workgroups: what a staged full-image render's 262 144-ray march uses) == rm_march_train_scan
    (one workgroup): the (id, offset, count) table and the totals, bit for bit, for ray counts around the workgroup size and at the largest size."""
    from nerf_signature_amd import _native as nv
    rng = np.random.RandomState(n % 1000)
    counts = torch.from_numpy(rng.randint(0, 1025, n).astype(np.int32)).cuda()
    counts[rng.randint(0, n, max(1, n // 7))] = 0                      # rays without samples share their successor's offset
    out = []
    for wide in (False, True):
        rays = torch.full((n, 3), -3, dtype=torch.int32, device="cuda")
        ctr = torch.full((2,), -3, dtype=torch.int32, device="cuda")
        if wide:
            sums = torch.full((int(nv.fn("rm_march_train_scan_blocks")(n)),), -77, dtype=torch.int32, device="cuda")
            nv.call("rm_march_train_scan_wide", nv.ptr(counts), n, nv.ptr(rays), nv.ptr(ctr), nv.ptr(sums), nv.stream())
        else:
            nv.call("rm_march_train_scan", nv.ptr(counts), n, nv.ptr(rays), nv.ptr(ctr), nv.stream())
        out.append((rays, ctr))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    c = counts.cpu().numpy().astype(np.int64)
    r = out[1][0].cpu().numpy().astype(np.int64)
    assert np.array_equal(r[:, 0], np.arange(n)) and np.array_equal(r[:, 2], c) and np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(c)[:-1]]))
    assert out[1][1].tolist() == [int(c.sum()), n]
