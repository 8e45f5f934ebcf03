"""Pins SURVEY.md 8(a) rows R2 / R3 / R8 / R10 / R11 to the REFERENCE ITSELF: the reference's own native extension
(/root/reference/raymarching/src/raymarching.cu + bindings.cpp, built for gfx950 by oracle/build_ref.py into
oracle/_ref/_raymarching_ref.so) runs on the GPU next to libnerfsig's rm_* entry points (through the C ABI) and next to
the C restatement oracle/raymarch_ref.c (on the CPU), on the same inputs.

Bars: integer results (per-ray sample counts, totals, morton codes, bitfields, alive lists) and every marched sample
(positions, directions, deltas) bit-exact; compositing within 2e-6 absolute (the reference accumulates serially per ray,
the product with wave-level prefix products/sums: a different fp32 summation order), compositing gradients within 1e-4
relative."""
import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr
from oracle import raymarch_ref as orm
from oracle import ref_native as ref

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref.available(), reason="oracle/_ref/_raymarching_ref.so not built (python -m oracle.build_ref)")]


@pytest.fixture(scope="module")
def rmod():
    from nerf_signature_amd import raymarching
    return raymarching


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _np(t):
    return t.detach().cpu().numpy()


def _scene(n, bound, seed=0, radius=None):
    grid, bitfield, C = cf.ball_scene(bound=bound)
    pose, intr, inds = cf.orbit_rays(n, seed=seed, radius=radius or (3.2248 if bound == 1.0 else 1.3))
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    return grid, bitfield, C, o[0].contiguous().numpy(), d[0].contiguous().numpy()


def _by_ray_id(rays):
    """The reference stores (ray id, offset, count) at an atomic-arrival slot: bring the table into ray-id order."""
    rays = rays.long()
    order = torch.argsort(rays[:, 0])
    rays = rays[order]
    assert torch.equal(rays[:, 0], torch.arange(rays.shape[0], device=rays.device)), "every ray id appears exactly once"
    return rays


def _gather_points(rays_sorted, *buffers):
    """Rows of the reference's point buffers re-ordered into ray-id order (the product's canonical layout)."""
    counts = rays_sorted[:, 2]
    start = torch.repeat_interleave(rays_sorted[:, 1], counts)
    first = torch.repeat_interleave(torch.cumsum(counts, 0) - counts, counts)
    idx = start + (torch.arange(int(counts.sum()), device=counts.device) - first)
    return [b[idx] for b in buffers]


def test_reference_module_exports_the_ten_entry_points():
    names = {"near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train", "composite_rays_train_forward",
             "composite_rays_train_backward", "march_rays", "composite_rays"}
    assert names <= set(dir(ref.module()))           # raymarching/src/bindings.cpp:5-18


def test_morton_packbits_vs_reference(rmod):
    rng = np.random.RandomState(0)
    c = rng.randint(0, 128, size=(10007, 3)).astype(np.int32)
    want = ref.morton3D(_cuda(c))
    assert torch.equal(rmod.morton3D(_cuda(c)), want)
    np.testing.assert_array_equal(orm.morton3D(c), _np(want))
    back = ref.morton3D_invert(want)
    assert torch.equal(rmod.morton3D_invert(want), back) and torch.equal(back.cpu(), torch.from_numpy(c))
    np.testing.assert_array_equal(orm.morton3D_invert(_np(want)), _np(back))
    for grid, thresh in ((rng.randn(2, 128 ** 3 // 16).astype(np.float32), 0.1), (cf.ball_scene(bound=1.0)[0], 10.0), (cf.ball_scene(bound=2.0)[0], 0.01)):
        want = ref.packbits(_cuda(grid), thresh)
        assert torch.equal(rmod.packbits(_cuda(grid), thresh), want)
        np.testing.assert_array_equal(orm.packbits(grid, thresh), _np(want))


@pytest.mark.parametrize("bound", [1.0, 2.0, 8.0])
def test_near_far_vs_reference(rmod, bound):
    _, _, _, o, d = _scene(4096, bound)
    d[5] = (0, 1, 0)     # axis-parallel ray (infinite reciprocal)
    o[6] = (9, 9, 9)     # misses the box
    o[7], d[7] = (0.1, 0.2, -0.3), (0.6, 0.0, 0.8)     # starts inside
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    n_ref, f_ref = ref.near_far_from_aabb(_cuda(o), _cuda(d), _cuda(aabb), 0.2)
    n1, f1 = rmod.near_far_from_aabb(_cuda(o), _cuda(d), _cuda(aabb), 0.2)
    assert torch.equal(n1, n_ref) and torch.equal(f1, f_ref)
    with np.errstate(divide="ignore", invalid="ignore"):
        n0, f0 = orm.near_far_from_aabb(o, d, aabb, 0.2)
    np.testing.assert_array_equal(n0, _np(n_ref))
    np.testing.assert_array_equal(f0, _np(f_ref))


def test_sph_from_ray_vs_reference(rmod):
    _, _, _, o, d = _scene(1024, 1.0)
    want = ref.sph_from_ray(_cuda(o * 0.1), _cuda(d), 4.0)
    np.testing.assert_allclose(_np(rmod.sph_from_ray(_cuda(o * 0.1), _cuda(d), 4.0)), _np(want), rtol=0, atol=2e-6)
    np.testing.assert_allclose(orm.sph_from_ray(o * 0.1, d, 4.0), _np(want), rtol=0, atol=2e-6)


MARCH_CASES = [(1.0, 0.0, 4096, 1024, False), (2.0, 0.0, 2048, 1024, False), (2.0, 1 / 128, 2048, 1024, False), (1.0, 0.0, 77, 1024, False),
               (1.0, 0.0, 2048, 1024, True), (2.0, 1 / 128, 1024, 1024, True), (1.0, 0.0, 1024, 256, False), (1.0, 0.0, 512, 64, False),
               # more cascades (bound 8 -> 4, bound 16 -> 5: the 360-degree scenes' settings), a bound that is not a power of two
               (8.0, 1 / 128, 1024, 1024, False), (16.0, 1 / 256, 1024, 1024, True), (1.5, 0.0, 1024, 1024, False)]


@pytest.mark.parametrize("bound,dt_gamma,n,max_steps,perturb", MARCH_CASES)
def test_march_train_vs_reference(rmod, bound, dt_gamma, n, max_steps, perturb):
    """kernel_march_rays_train (raymarching.cu:312-480) against rm_march_train_* and against oracle_march_rays_train:
    per-ray counts, totals and every sample bit-exact.  max_steps 64 is the dt_min > dt_max corner."""
    _, bitfield, C, o, d = _scene(n, bound)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    oc, dc, bf = _cuda(o), _cuda(d), _cuda(bitfield)
    nears, fars = ref.near_far_from_aabb(oc, dc, _cuda(aabb), 0.2)
    noises = torch.from_numpy(np.random.RandomState(1).rand(n).astype(np.float32)).cuda() if perturb else None
    x_r, d_r, dl_r, rays_r, ctr_r = ref.march_rays_train(oc, dc, bound, bf, C, 128, nears, fars, noises, dt_gamma, max_steps)
    torch.cuda.synchronize()
    rays_r = _by_ray_id(rays_r)
    total = int(ctr_r[0])
    assert int(ctr_r[1]) == n and int(rays_r[:, 2].sum()) == total and total > 0
    x_r, d_r, dl_r = _gather_points(rays_r, x_r, d_r, dl_r)

    # product, through the C ABI (counts + scan + write; deterministic ray-id order)
    ctr1 = torch.zeros(2, dtype=torch.int32, device="cuda")
    zero = torch.zeros(n, device="cuda")
    _, _, rays1, write = rmod.march_rays_train_device(oc, dc, bound, bf, C, 128, nears, fars, ctr1, noises if perturb else zero, dt_gamma, max_steps)
    assert torch.equal(ctr1.cpu(), ctr_r.cpu())
    assert torch.equal(rays1[:, 2].long(), rays_r[:, 2]), "per-ray sample counts differ from the reference kernel"
    x1, d1, dl1 = write(total)
    assert torch.equal(x1[:total], x_r) and torch.equal(d1[:total], d_r) and torch.equal(dl1[:total], dl_r)

    # the C restatement, on the CPU
    ctr0 = np.zeros(2, np.int32)
    x0, d0, dl0, rays0 = orm.march_rays_train(o, d, bound, bitfield, C, 128, _np(nears), _np(fars), ctr0, -1, perturb, -1, True, dt_gamma, max_steps,
                                              noises=None if noises is None else _np(noises))
    np.testing.assert_array_equal(ctr0, _np(ctr_r))
    np.testing.assert_array_equal(rays0[:, 2], _np(rays_r[:, 2]))
    np.testing.assert_array_equal(x0[:total], _np(x_r))
    np.testing.assert_array_equal(d0[:total], _np(d_r))
    np.testing.assert_array_equal(dl0[:total], _np(dl_r))


@pytest.mark.parametrize("scene", ["hotdog", "counter"])
def test_bench_workload_march_vs_reference(rmod, scene):
    """All 4608 block rays + 4096 content rays of the bench step -- scene S0 (hotdog-like, one cascade) and scene S1 (counter-like: bound 2,
    two cascades, camera inside, ~500 / ~240 samples per ray, the cascade chosen per sample by mip_from_pos / mip_from_dt,
    raymarching.cu:42-54,368) -- counts and every sample bit-exact against the reference's own kernel."""
    from nerf_signature_amd import synthetic
    cfg = synthetic.SCENES[scene]
    b = cfg["bound"]
    C = 1 if b == 1.0 else 2
    bits = torch.from_numpy(synthetic.pack_bits_np(synthetic.density_grid(b), 10.0)[0]).cuda()
    aabb = torch.tensor([-b, -b, -b, b, b, b], device="cuda")
    bo, bd = synthetic.block_rays(scene, "cuda")
    co, cd = synthetic.content_rays(scene, 4096, seed=0, device="cuda")
    for o, d in ((bo.reshape(-1, 3), bd.reshape(-1, 3)), (co.reshape(-1, 3), cd.reshape(-1, 3))):
        o, d = o.contiguous(), d.contiguous()
        n = o.shape[0]
        nears, fars = ref.near_far_from_aabb(o, d, aabb, 0.2)
        x_r, d_r, dl_r, rays_r, ctr_r = ref.march_rays_train(o, d, b, bits, C, 128, nears, fars, dt_gamma=cfg["dt_gamma"])
        rays_r = _by_ray_id(rays_r)
        total = int(ctr_r[0])
        x_r, dl_r = _gather_points(rays_r, x_r, dl_r)
        ctr1 = torch.zeros(2, dtype=torch.int32, device="cuda")
        x1, d1, dl1, rays1 = rmod.march_rays_train(o, d, b, bits, C, 128, nears, fars, ctr1, -1, False, 128, True, cfg["dt_gamma"], 1024)
        assert int(ctr1[0]) == total and torch.equal(rays1[:, 2].long(), rays_r[:, 2])
        assert total > (100_000 if scene == "hotdog" else 900_000)
        assert torch.equal(x1[:total], x_r) and torch.equal(dl1[:total], dl_r)
        assert not bool(x1[total:].any())                    # alignment padding is zero rows (raymarching.py:205-207,225-229)


def test_composite_train_vs_reference(rmod):
    """kernel_composite_rays_train_forward/backward (raymarching.cu:501-693) on the reference's own (atomic-order) ray table."""
    _, bitfield, C, o, d = _scene(2048, 1.0)
    oc, dc, bf = _cuda(o), _cuda(d), _cuda(bitfield)
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device="cuda")
    nears, fars = ref.near_far_from_aabb(oc, dc, aabb, 0.2)
    _, _, dl_full, rays_atomic, ctr = ref.march_rays_train(oc, dc, 1.0, bf, C, 128, nears, fars)
    M = int(ctr[0]) + 128 - int(ctr[0]) % 128
    deltas = dl_full[:M].contiguous()
    rng = np.random.RandomState(4)
    for scale in (2.0, 300.0):   # thin medium, and opaque enough to hit the T < 1e-4 early exit
        sig, rgb = _cuda((rng.rand(M) * scale).astype(np.float32)), _cuda(rng.rand(M, 3).astype(np.float32))
        g_ws, g_img = _cuda(rng.randn(2048).astype(np.float32)), _cuda(rng.randn(2048, 3).astype(np.float32))
        ws_r, dep_r, img_r = ref.composite_rays_train_forward(sig, rgb, deltas, rays_atomic, 1e-4)
        gs_r, gc_r = ref.composite_rays_train_backward(g_ws, g_img, sig, rgb, deltas, rays_atomic, ws_r, img_r, 1e-4)
        # product: the same table (any slot order is legal input: outputs are indexed by ray id)
        s1, c1 = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
        ws1, dep1, img1 = rmod.composite_rays_train(s1, c1, deltas, rays_atomic, 1e-4)
        np.testing.assert_allclose(_np(ws1), _np(ws_r), rtol=0, atol=2e-6)
        np.testing.assert_allclose(_np(dep1), _np(dep_r), rtol=0, atol=1e-5)
        np.testing.assert_allclose(_np(img1), _np(img_r), rtol=0, atol=2e-6)
        torch.autograd.backward([ws1, img1], [g_ws, g_img])
        np.testing.assert_allclose(_np(c1.grad), _np(gc_r), rtol=0, atol=2e-6)
        np.testing.assert_allclose(_np(s1.grad), _np(gs_r), rtol=1e-4, atol=2e-6 * max(1.0, 3.0 / scale))
        # the C restatement
        ws0, dep0, img0 = orm.composite_rays_train_forward(_np(sig), _np(rgb), _np(deltas), _np(rays_atomic), 1e-4)
        np.testing.assert_allclose(ws0, _np(ws_r), rtol=0, atol=2e-6)
        np.testing.assert_allclose(dep0, _np(dep_r), rtol=0, atol=1e-5)
        np.testing.assert_allclose(img0, _np(img_r), rtol=0, atol=2e-6)
        gs0, gc0 = orm.composite_rays_train_backward(_np(g_ws), _np(g_img), _np(sig), _np(rgb), _np(deltas), _np(rays_atomic), ws0, img0, 1e-4)
        np.testing.assert_allclose(gc0, _np(gc_r), rtol=0, atol=2e-6)
        np.testing.assert_allclose(gs0, _np(gs_r), rtol=1e-4, atol=2e-6 * max(1.0, 3.0 / scale))


@pytest.mark.parametrize("bound,dt_gamma", [(1.0, 0.0), (2.0, 1 / 128), (8.0, 1 / 128)])
def test_eval_loop_vs_reference(rmod, bound, dt_gamma):
    """kernel_march_rays / kernel_composite_rays (raymarching.cu:701-914) driven like renderer_wtmk.py:335-367, reference and product
    side by side: sample bursts and alive lists bit-exact every round, accumulated outputs within summation-order tolerance."""
    _, bitfield, C, o, d = _scene(1000, bound)
    N = 1000
    oc, dc, bf = _cuda(o), _cuda(d), _cuda(bitfield)
    aabb = _cuda(np.array([-bound] * 3 + [bound] * 3, np.float32))
    nears, fars = ref.near_far_from_aabb(oc, dc, aabb, 0.2)
    sig_of = lambda p: 40.0 * (0.5 + p[:, 0] / bound)
    rgb_of = lambda p: torch.stack([0.5 + 0.4 * p[:, 1] / bound, 0.3 + 0 * p[:, 1], 0.5 - 0.4 * p[:, 2] / bound], -1).contiguous()
    acc_r = (torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, 3, device="cuda"))
    acc_1 = tuple(torch.zeros_like(t) for t in acc_r)
    alive_r, t_r = torch.arange(N, dtype=torch.int32, device="cuda"), nears.clone()
    alive_1, t_1 = alive_r.clone(), nears.clone()
    # and the C restatement
    acc_0 = (np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32))
    alive_0, t_0 = np.arange(N, dtype=np.int32), _np(nears).copy()
    step, rounds = 0, 0
    while step < 1024 and alive_r.shape[0] > 0:
        n_alive = alive_r.shape[0]
        n_step = max(min(N // n_alive, 8), 1)
        p_r, dd_r, dl_r = ref.march_rays(n_alive, n_step, alive_r, t_r, oc, dc, bound, bf, C, 128, nears, fars, 128, None, dt_gamma, 1024)
        p_1, dd_1, dl_1 = rmod.march_rays(n_alive, n_step, alive_1, t_1, oc, dc, bound, bf, C, 128, nears, fars, 128, False, dt_gamma, 1024)
        assert torch.equal(p_1, p_r) and torch.equal(dd_1, dd_r) and torch.equal(dl_1, dl_r)
        p_0, dd_0, dl_0 = orm.march_rays(n_alive, n_step, alive_0, t_0, o, d, bound, bitfield, C, 128, _np(nears), _np(fars), 128, False, dt_gamma, 1024)
        np.testing.assert_array_equal(p_0, _np(p_r))
        np.testing.assert_array_equal(dl_0, _np(dl_r))
        sig, rgb = sig_of(p_r).contiguous(), rgb_of(p_r)
        ref.composite_rays(n_alive, n_step, alive_r, t_r, sig, rgb, dl_r, *acc_r, 1e-2)
        rmod.composite_rays(n_alive, n_step, alive_1, t_1, sig, rgb, dl_1, *acc_1, 1e-2)
        orm.composite_rays(n_alive, n_step, alive_0, t_0, _np(sig), _np(rgb), dl_0, *acc_0, 1e-2)
        assert torch.equal(alive_1, alive_r), "terminated-ray flags differ from the reference kernel"
        assert torch.equal(t_1, t_r)
        np.testing.assert_array_equal(alive_0, _np(alive_r))
        alive_r = alive_r[alive_r >= 0].contiguous()                      # the reference's host-side compaction (renderer_wtmk.py:363)
        out, n_out = rmod.compact_alive(alive_1)
        assert int(n_out.item()) == alive_r.shape[0]
        alive_1 = out[:alive_r.shape[0]].contiguous()
        assert torch.equal(alive_1, alive_r)
        alive_0 = np.ascontiguousarray(alive_0[alive_0 >= 0])
        step += n_step
        rounds += 1
    assert rounds > 3
    for a1, ar, a0, tol in zip(acc_1, acc_r, acc_0, (5e-6, 2e-5, 5e-6)):
        np.testing.assert_allclose(_np(a1), _np(ar), rtol=0, atol=tol)
        np.testing.assert_allclose(a0, _np(ar), rtol=0, atol=tol)
