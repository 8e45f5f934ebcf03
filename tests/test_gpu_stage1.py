"""N3 (stage-1 support): gradients of every parameter of the clean model (all 16 base tables, both MLPs) against the oracle's
autograd; the weight-gradient kernel against plain matrix products; the device-row-count entry points; the captured loop against
the eager loop, against the CPU oracle over a 200-step trajectory, with the density-grid refresh in its cadence, and with two
data-parallel ranks against the single-process gradient."""
import copy
import threading

import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr

pytestmark = pytest.mark.gpu
KW = dict(dt_gamma=0, max_steps=1024)


def _clean_model(bound=1.0, mlp_scale=1.0):
    from nerf_signature_amd.stage1 import CleanNeRFNetwork
    m = CleanNeRFNetwork(bound=bound, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
    grid, bitfield, C = cf.ball_scene(bound=bound)
    with torch.no_grad():
        for l in range(16):
            m.encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(l)))
        m.sigma_net.params.copy_(torch.from_numpy(cf.mlp_params(3072, 1337)) * mlp_scale)
        m.color_net.params.copy_(torch.from_numpy(cf.mlp_params(7168, 1338)) * mlp_scale)
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bitfield))
    return m.cuda().train(), bitfield, C


def _oracle_params(m):
    return {"bound": float(m.bound), "base_tables": [e.weight.detach().cpu().clone().requires_grad_(True) for e in m.encoder.embeddings],
            "cb_tables": [], "sigma_params": m.sigma_net.params.detach().cpu().clone().requires_grad_(True),
            "color_params": m.color_net.params.detach().cpu().clone().requires_grad_(True)}


def _scene(m, bitfield):
    return {"bound": float(m.bound), "cascade": m.cascade, "grid_size": 128, "density_bitfield": np.ascontiguousarray(bitfield),
            "aabb": np.array([-m.bound] * 3 + [m.bound] * 3, np.float32), "min_near": 0.2, "density_scale": 1}


def _patch_rays(n_side=8, seed=2, lo=184):
    """n_side x n_side pixels through the middle of the ball (orbit camera of the bench scene)."""
    pose, intr, _ = cf.orbit_rays(1, seed=seed)
    rr, cc = np.meshgrid(np.arange(lo, lo + n_side), np.arange(lo, lo + n_side), indexing="ij")
    inds = torch.from_numpy((rr * 400 + cc).reshape(-1).astype(np.int64))
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, inds[None])
    return o[0].contiguous(), d[0].contiguous()


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("bound", [1.0, 2.0], ids=["one_cascade", "two_cascades"])
def test_all_parameter_gradients_vs_oracle(bound):
    """bound = 2: the config-3-like scene S1 (two cascades); the field sees positions in [-2, 2] scaled into the encoder's unit box (network_hash.py:97-99)."""
    m, bitfield, C = _clean_model(bound=bound)
    assert C == (1 if bound == 1.0 else 2) and m.cascade == C
    trainable = {n for n, p in m.named_parameters() if p.requires_grad and p.numel()}
    assert trainable == {f"encoder.embeddings.{l}.weight" for l in range(16)} | {"sigma_net.params", "color_net.params"}
    rng = np.random.RandomState(0)
    M = 3001
    pts = torch.from_numpy(((rng.rand(M, 3) * 2 - 1) * bound).astype(np.float32))
    pts[:3] = torch.tensor([[-1., -1, -1], [1, 1, 1], [0.5, -0.25, 0]]) * bound
    dirs = torch.from_numpy(cf.unit_dirs(M, seed=9))
    gs = torch.from_numpy(rng.randn(M).astype(np.float32))
    gc = torch.from_numpy(rng.randn(M, 3).astype(np.float32))
    P = _oracle_params(m)
    s0, c0 = fr.field_forward(pts, dirs, None, P)
    ((s0 * gs).sum() + (c0 * gc).sum()).backward()
    s1, c1 = m(pts.cuda(), dirs.cuda())
    np.testing.assert_allclose(s1.detach().cpu().numpy(), s0.detach().numpy(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(c1.detach().cpu().numpy(), c0.detach().numpy(), rtol=0, atol=1e-3)
    ((s1 * gs.cuda()).sum() + (c1 * gc.cuda()).sum()).backward()
    # MLP weight gradients (sums over all points: ReLU-boundary flips average out)
    assert rel(m.sigma_net.params.grad, P["sigma_params"].grad) < 2e-3
    assert rel(m.color_net.params.grad, P["color_params"].grad) < 2e-3
    assert float(m.color_net.params.grad[6144 + 3 * 64:].abs().max()) == 0.0        # rows 3..15 of the padded colour head: no gradient
    worst = 0.0
    for l in range(16):          # every level: the same rows touched, the same values
        g1, g0 = m.encoder.embeddings[l].weight.grad.cpu(), P["base_tables"][l].grad
        assert torch.equal(g1 != 0, g0 != 0) or float(((g1 != 0) != (g0 != 0)).float().mean()) < 1e-4, l
        worst = max(worst, rel(g1, g0))
        assert rel(g1, g0) < 5e-3, l
    print(f"\nworst base-table gradient rel. L2 over the 16 levels: {worst:.2e}")


def _traced_pass(m, pts, dirs, gs, gc, capacity=None, rows=None, fused=True, half=False):
    """forward + backward through the explicit entry points (fused: field_bwd_wgrad; else field_bwd_trace + field_wgrad); returns (traces, sigma
    gradient, colour gradient, table gradients [16,T,2])."""
    from nerf_signature_amd import _native as nv, fieldops as fo, stage1
    M = pts.shape[0] if capacity is None else capacity
    dev = pts.device
    tr = stage1._Traces(M, dev, fused=fused, half=half)
    if capacity is not None:     # poison everything past the live rows: none of it may be read
        for t in [tr.planes, *tr.act, *(tr.d or []), tr.d_planes]:
            t.fill_(float("nan"))
    packed = fo.pack_weights(m.sigma_net.params, m.color_net.params)
    base_ptrs = nv.ptr_array([t.detach() for t in m.encoder.tables()])
    stage1._forward_trace(tr, pts, dirs, m.bound, base_ptrs, packed, rows=rows)
    g_sp, g_cp = torch.empty(3072, device=dev), torch.empty(7168, device=dev)
    stage1._backward_trace(tr, gs, gc, packed, g_sp, g_cp, rows=rows)
    plan = torch.empty(int(nv.fn("hg_levels_plan_bytes")(M)), dtype=torch.uint8, device=dev)
    nv.call("hg_levels_plan", nv.ptr(pts), M, nv.ptr(rows), float(m.bound), nv.ptr(plan), nv.stream())
    G = torch.full((16, 1 << 19, 2), float("nan"), device=dev)
    nv.call("hg_levels_scatter", nv.ptr(pts), M, nv.ptr(rows), float(m.bound), nv.ptr(tr.d_planes), tr.stride, nv.ptr(plan), nv.ptr_array([G[l] for l in range(16)]), nv.stream())
    return tr, g_sp, g_cp, G


def _random_batch(M, seed=0, dev="cuda"):
    rng = np.random.RandomState(seed)
    pts = torch.from_numpy((rng.rand(M, 3) * 2 - 1).astype(np.float32)).to(dev)
    dirs = torch.from_numpy(cf.unit_dirs(M, seed=9 + seed)).to(dev)
    gs = torch.from_numpy((rng.randn(M) * 1e-4).astype(np.float32)).to(dev)      # the size an unscaled MSE seed has
    gc = torch.from_numpy((rng.randn(M, 3) * 1e-4).astype(np.float32)).to(dev)
    return pts, dirs, gs, gc


def _fused_equals_two_launches(m, pts, dirs, gs, gc, tr, g_sp, g_cp, G, want_s, want_c):
    """field_bwd_wgrad against the pair it replaces (whose traces are `tr`): the same backward chain -> d_planes and the tables bit for bit; the weight
    gradients against the same float64 products (another order of the partial sums); twice the same bits."""
    M = pts.shape[0]
    trf, f_sp, f_cp, Gf = _traced_pass(m, pts, dirs, gs, gc, fused=True)
    assert trf.d is None
    assert torch.equal(trf.d_planes[:, :M], tr.d_planes[:, :M]) and torch.equal(Gf, G)
    assert rel(f_sp, want_s) < 1e-5 and rel(f_cp, want_c) < 1e-5, (rel(f_sp, want_s), rel(f_cp, want_c))
    assert rel(f_sp, g_sp) < 1e-5 and rel(f_cp, g_cp) < 1e-5, (rel(f_sp, g_sp), rel(f_cp, g_cp))      # two summation orders, each within 1e-5 of the float64 products
    np.testing.assert_allclose(f_cp.cpu().numpy(), want_c.float().cpu().numpy(), rtol=0, atol=2e-5 * float(want_c.abs().max()))
    assert float(f_cp[6144 + 3 * 64:].abs().max()) == 0.0                          # rows 3..15 of the padded colour head
    _, f_sp2, f_cp2, _ = _traced_pass(m, pts, dirs, gs, gc, fused=True)
    assert torch.equal(f_sp, f_sp2) and torch.equal(f_cp, f_cp2)


def test_weight_gradients_equal_the_matrix_products_and_are_bit_reproducible():
    """field_wgrad against float64 products of the very traces it reads; twice the same bits; the planned scatter against the record route;
    field_bwd_wgrad (the one-launch route, the default) against all of it."""
    from nerf_signature_amd import _native as nv
    m, _, _ = _clean_model()
    M = 20011
    pts, dirs, gs, gc = _random_batch(M)
    tr, g_sp, g_cp, G = _traced_pass(m, pts, dirs, gs, gc, fused=False)
    feat = tr.planes[:16, :M].double().permute(0, 2, 1).reshape(32, M)            # feature 2l + c
    hs, cin, h1, h2 = (a[:, :M].double() for a in tr.act)
    d_hs, d_so, d_h1, d_h2, d_out = (t[:, :M].double() for t in tr.d)
    want_s = torch.cat([(d_hs @ feat.t()).reshape(-1), (d_so @ hs.t()).reshape(-1)])
    want_c = torch.cat([(d_h1 @ cin.t()).reshape(-1), (d_h2 @ h1.t()).reshape(-1), (d_out @ h2.t()).reshape(-1)])
    assert rel(g_sp, want_s) < 1e-5 and rel(g_cp, want_c) < 1e-5, (rel(g_sp, want_s), rel(g_cp, want_c))      # split bf16 drops lo x lo: 2^-16 per product
    np.testing.assert_allclose(g_cp.cpu().numpy(), want_c.float().cpu().numpy(), rtol=0, atol=2e-5 * float(want_c.abs().max()))
    _, g_sp2, g_cp2, G2 = _traced_pass(m, pts, dirs, gs, gc, fused=False)
    assert torch.equal(g_sp, g_sp2) and torch.equal(g_cp, g_cp2) and torch.equal(G, G2)
    assert not torch.isnan(G).any()                                                # every row of every table written
    _fused_equals_two_launches(m, pts, dirs, gs, gc, tr, g_sp, g_cp, G, want_s, want_c)
    # the record route (hg_scatter_levels) computes the same sums: every contribution on its own there, runs of a cell pre-summed in fp32 on the
    # coarse levels here, and another fixed-point scale -- the same rows, values to fp32 rounding
    G3 = torch.empty_like(G)
    scratch = torch.empty(int(nv.fn("hg_scatter_levels_scratch_bytes")(M)), dtype=torch.uint8, device="cuda")
    nv.call("hg_scatter_levels", nv.ptr(pts), float(m.bound), nv.ptr(tr.d_planes), M, tr.stride, nv.ptr_array([G3[l] for l in range(16)]), nv.ptr(scratch), nv.stream())
    for l in range(16):
        assert torch.equal(G[l] != 0, G3[l] != 0), l
        assert rel(G[l], G3[l]) < 1e-6, (l, rel(G[l], G3[l]))


def test_half_precision_traces_round_only_what_is_saved():
    """trace_dtype="f16" (field_fwd_trace_f16 / field_bwd_wgrad_f16): the forward's outputs and ReLU masks, and the whole input-gradient chain (d_planes, the 16 table
    gradients), are the fp32 route's bit for bit -- nothing of them reads a saved layer input; the saved rows are the fp32 rows rounded to fp16 (the precision the
    reference's MLPs keep their activations in, tinycudann FullyFusedMLP); the weight gradients are the float64 products of THOSE rows (1e-5, as the fp32 route is of
    its rows) and sit within 3e-4 of the fp32 route's -- with a host and with a device row count, partial last tile included."""
    m, _, _ = _clean_model()
    for M, capacity in ((20011, None), (3000, 4000)):
        pts, dirs, gs, gc = _random_batch(M if capacity is None else capacity)
        rows = None if capacity is None else torch.tensor([M, 0], dtype=torch.int32, device="cuda")
        t32, s32, c32, G32 = _traced_pass(m, pts, dirs, gs, gc, capacity=capacity, rows=rows, fused=True)
        t16, s16, c16, G16 = _traced_pass(m, pts, dirs, gs, gc, capacity=capacity, rows=rows, fused=True, half=True)
        assert all(a.dtype == torch.float16 for a in t16.act) and sum(a.numel() * a.element_size() for a in t16.act) * 2 == sum(a.numel() * a.element_size() for a in t32.act)
        assert torch.equal(t16.sig[:M], t32.sig[:M]) and torch.equal(t16.rgb[:M], t32.rgb[:M]) and torch.equal(t16.masks[:(M + 31) // 32 * 32], t32.masks[:(M + 31) // 32 * 32])
        for a16, a32 in zip(t16.act, t32.act):
            # round to nearest even, element by element.  (The SH inputs are products the compiler rounds ONCE to fp16 -- v_fma_mixlo_f16 -- where fp32-then-fp16
            # rounds twice: on an exact tie of the fp32 value the two differ by one fp16 ulp; measured 1 element of 640 352.)
            want = a32[:, :M].half()
            off = a16[:, :M] != want
            assert float(off.float().mean()) < 1e-5, float(off.float().mean())
            if off.any():
                ulp = torch.maximum(want.float().abs(), torch.tensor(2.0 ** -14, device="cuda")) * 2.0 ** -10
                assert bool(((a16[:, :M].float() - want.float()).abs()[off] <= ulp[off]).all())
        assert torch.equal(t16.d_planes[:, :M], t32.d_planes[:, :M]) and torch.equal(G16, G32)
        if capacity is None:
            # the pre-activation gradients of the chain (from the two-launch route: the same chain) x the ROUNDED layer inputs, in float64
            tr2, _, _, _ = _traced_pass(m, pts, dirs, gs, gc, fused=False)
            feat = tr2.planes[:16, :M].double().permute(0, 2, 1).reshape(32, M)
            hs, cin, h1, h2 = (a[:, :M].double() for a in t16.act)
            d_hs, d_so, d_h1, d_h2, d_out = (t[:, :M].double() for t in tr2.d)
            want_s = torch.cat([(d_hs @ feat.t()).reshape(-1), (d_so @ hs.t()).reshape(-1)])
            want_c = torch.cat([(d_h1 @ cin.t()).reshape(-1), (d_h2 @ h1.t()).reshape(-1), (d_out @ h2.t()).reshape(-1)])
            assert rel(s16, want_s) < 1e-5 and rel(c16, want_c) < 1e-5, (rel(s16, want_s), rel(c16, want_c))
        print(f"\nfp16 traces, {M} points: weight gradients vs the fp32 traces' rel. L2 sigma MLP {rel(s16, s32):.2e}, colour MLP {rel(c16, c32):.2e}")
        assert rel(s16, s32) < 3e-4 and rel(c16, c32) < 3e-4
        assert float(c16[6144 + 3 * 64:].abs().max()) == 0.0


def test_pipelined_trace_forward_equals_the_plain_loop_bit_for_bit():
    """field_fwd_trace's launch (k_field_fwd_trace: the next tile's inputs requested in front of a tile's trace stores) against the generic kernel's trace variant
    (mlp_set_pipelined bit 0 clear): outputs, ReLU masks and every saved layer input bit for bit -- with a host and with a device row count."""
    from nerf_signature_amd import _native as nv, fieldops as fo, stage1
    m, _, _ = _clean_model()
    packed = fo.pack_weights(m.sigma_net.params, m.color_net.params)
    base_ptrs = nv.ptr_array([t.detach() for t in m.encoder.tables()])
    before = nv.fn("mlp_get_pipelined")()
    try:
        for M, n in ((20011, None), (21000, 7013), (33, None), (4096, 1)):
            pts, dirs, _, _ = _random_batch(M, seed=2)
            rows = None if n is None else torch.tensor([n, 0], dtype=torch.int32, device="cuda")
            live = M if n is None else n
            got = []
            for mask in (3, 2):
                nv.call("mlp_set_pipelined", mask)
                tr = stage1._Traces(M, pts.device, with_grads=False)
                for t in [tr.sig, tr.rgb, *tr.act]:
                    t.fill_(float("nan"))
                tr.masks.fill_(-1)
                stage1._forward_trace(tr, pts, dirs, m.bound, base_ptrs, packed, rows=rows)
                got.append(tr)
            a, b = got
            assert torch.equal(a.sig[:live], b.sig[:live]) and torch.equal(a.rgb[:live], b.rgb[:live]) and not torch.isnan(a.sig[:live]).any()
            n_tiles = (live + 31) // 32
            assert torch.equal(a.masks.view(-1)[:n_tiles * 192], b.masks.view(-1)[:n_tiles * 192])
            for x, y in zip(a.act, b.act):
                assert torch.equal(x[:, :live], y[:, :live]) and not torch.isnan(x[:, :live]).any()
            if n is not None and n_tiles * 32 < M:      # nothing past the live tiles was written
                assert torch.isnan(a.sig[n_tiles * 32:]).all() and torch.isnan(a.act[0][:, n_tiles * 32:]).all() and bool((a.masks.view(-1)[n_tiles * 192:] == -1).all())
    finally:
        nv.call("mlp_set_pipelined", before)


@pytest.mark.parametrize("fused", [True, False], ids=["one_launch", "two_launches"])
def test_device_row_count_entry_points_walk_only_the_live_rows(fused):
    """Buffers of capacity 3 x the live rows, NaN past them: tables bit for bit, MLP gradients to summation order (the split over workgroups follows the capacity)."""
    m, _, _ = _clean_model()
    n, cap = 7013, 21000
    pts, dirs, gs, gc = _random_batch(cap, seed=1)
    _, g_sp0, g_cp0, G0 = _traced_pass(m, pts[:n].contiguous(), dirs[:n].contiguous(), gs[:n].contiguous(), gc[:n].contiguous(), fused=fused)
    pts[n:], dirs[n:], gs[n:], gc[n:] = float("nan"), float("nan"), float("nan"), float("nan")
    rows = torch.tensor([n, 0], dtype=torch.int32, device="cuda")
    tr, g_sp1, g_cp1, G1 = _traced_pass(m, pts, dirs, gs, gc, capacity=cap, rows=rows, fused=fused)
    assert torch.equal(G0, G1)
    assert rel(g_sp1, g_sp0) < 1e-6 and rel(g_cp1, g_cp0) < 1e-6
    assert torch.isnan(tr.d_planes[0, cap - 1]).all()                             # the tail was never touched
    rows.zero_()                                                                  # no live row at all: zero gradients, nothing read
    _, g_sp2, g_cp2, G2 = _traced_pass(m, pts, dirs, gs, gc, capacity=cap, rows=rows, fused=fused)
    assert float(G2.abs().max()) == 0.0 and float(g_sp2.abs().max()) == 0.0 and float(g_cp2.abs().max()) == 0.0


def test_bench_size_step_properties():
    """VERDICT round 4, weak #8 ("nothing at bench size"): the samples a 4096-ray step of the bench scene really marches (a ray's consecutive samples share cells on the coarse
    levels: the merged-run path, skewed slices) -- once through the synthetic grid (~140 k points) and once through a grid filled everywhere (what update_extra_state makes of a
    random field: ~600 k points) -- checked through properties that need no oracle pass of that size: weight gradients against float64 products of the traces the kernel read,
    the planned table scatter against the record route (every contribution on its own) level by level, bit-reproducibility, linearity of both in the upstream gradient."""
    from nerf_signature_amd import _native as nv, synthetic
    from nerf_signature_amd.stage1 import CleanNeRFNetwork
    m = CleanNeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
    with torch.no_grad():
        for l, e in enumerate(m.encoder.embeddings):
            e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
        grid = synthetic.density_grid(1.0)
        bits, _ = synthetic.pack_bits_np(grid, 10.0)
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bits))
    m = m.cuda().train()
    o, d = synthetic.content_rays("hotdog", 4096, 0, torch.device("cuda"))
    for full_grid in (False, True):
        if full_grid:
            m.density_bitfield.fill_(255)
        rec = m.march_ahead(o, d, 0, 1024, perturb=False, capacity=128)       # a counting march first: the exact point total
        n = int(rec["counter"][0])
        m.drop_marched()
        from nerf_signature_amd.raymarching import padded_point_count
        rec = m.march_ahead(o, d, 0, 1024, perturb=False, capacity=padded_point_count(n))
        assert int(rec["counter"][0]) == n and (n > 500_000 if full_grid else 100_000 < n < 200_000), n
        pts, dirs = rec["xyzs"][:n].contiguous(), rec["dirs"][:n].contiguous()
        rng = np.random.RandomState(3)
        gs = torch.from_numpy((rng.randn(n) * 1e-4).astype(np.float32)).cuda()
        gc = torch.from_numpy((rng.randn(n, 3) * 1e-4).astype(np.float32)).cuda()
        tr, g_sp, g_cp, G = _traced_pass(m, pts, dirs, gs, gc, fused=False)
        feat = tr.planes[:16, :n].double().permute(0, 2, 1).reshape(32, n)
        hs, cin, h1, h2 = (a[:, :n].double() for a in tr.act)
        d_hs, d_so, d_h1, d_h2, d_out = (t[:, :n].double() for t in tr.d)
        want_s = torch.cat([(d_hs @ feat.t()).reshape(-1), (d_so @ hs.t()).reshape(-1)])
        want_c = torch.cat([(d_h1 @ cin.t()).reshape(-1), (d_h2 @ h1.t()).reshape(-1), (d_out @ h2.t()).reshape(-1)])
        assert rel(g_sp, want_s) < 1e-5 and rel(g_cp, want_c) < 1e-5, (n, rel(g_sp, want_s), rel(g_cp, want_c))
        assert not torch.isnan(G).any()
        G3 = torch.empty_like(G)
        scratch = torch.empty(int(nv.fn("hg_scatter_levels_scratch_bytes")(n)), dtype=torch.uint8, device="cuda")
        nv.call("hg_scatter_levels", nv.ptr(pts), float(m.bound), nv.ptr(tr.d_planes), n, tr.stride, nv.ptr_array([G3[l] for l in range(16)]), nv.ptr(scratch), nv.stream())
        for l in range(16):
            assert torch.equal(G[l] != 0, G3[l] != 0), (n, l)
            assert rel(G[l], G3[l]) < 2e-6, (n, l, rel(G[l], G3[l]))
        _, g_sp2, g_cp2, G2 = _traced_pass(m, pts, dirs, gs, gc, fused=False)
        assert torch.equal(g_sp, g_sp2) and torch.equal(g_cp, g_cp2) and torch.equal(G, G2)
        _fused_equals_two_launches(m, pts, dirs, gs, gc, tr, g_sp, g_cp, G, want_s, want_c)            # the one-launch route (the default) at this size
        _, f_sp, f_cp, _ = _traced_pass(m, pts, dirs, gs, gc)
        _, g_sp4, g_cp4, G4 = _traced_pass(m, pts, dirs, gs * 4, gc * 4)                    # a power of two: exact in every float operation of the chain
        assert torch.equal(g_sp4, f_sp * 4) and torch.equal(g_cp4, f_cp * 4) and rel(G4, G * 4) < 1e-6
        m.drop_marched()
        del tr, G, G2, G3, G4


def _adam(m, lr=1e-2):
    return torch.optim.Adam(m.get_params(lr), betas=(0.9, 0.99), eps=1e-15)


def test_captured_loop_equals_the_eager_loop():
    """Five optimisation steps on fixed rays: the captured explicit-kernel step against the autograd step with torch's own Adam."""
    from nerf_signature_amd.stage1 import CleanLoop, GraphedCleanLoop
    o, d = _patch_rays(16)
    target = torch.tensor([0.2, 0.5, 0.8]).view(1, 3).expand(256, 3).contiguous().cuda()
    data = {"rays_o": o.cuda()[None], "rays_d": d.cuda()[None], "images": target[None], "perturb": False, "force_all_rays": True}
    m0, _, _ = _clean_model()
    eager = CleanLoop(m0, _adam(m0), KW, update_extra_interval=10 ** 9)
    eager.global_step = 1
    l0 = [float(eager.step(data)[1].detach()) for _ in range(5)]
    m1, _, _ = _clean_model()
    loop = GraphedCleanLoop(m1, _adam(m1), KW, n_rays=256, update_extra_interval=0, perturb=False)
    l1 = [float(loop.step(data)) for _ in range(5)]
    assert len(loop.graph.segments) == 1 and not loop.overflowed()
    np.testing.assert_allclose(l1, l0, rtol=2e-4)
    assert l0[-1] < 0.7 * l0[0]
    # Adam with eps = 1e-15 turns a gradient whose SIGN is rounding noise into +-lr per step (lr = 1e-2): the two loops size their launches differently --
    # exact point count against capacity -- so their weight-gradient slabs are cut differently (same sums, other rounding), the MLP weights drift apart by ~1e-5
    # and a few table entries whose gradients nearly cancel (the fine levels: 1-2 in 10^4) take the other sign for a step or two.  Both loops are bit-reproducible
    # by themselves (tools/stage1_determinism.py).
    for i, (a, b) in enumerate(zip(m1.trainable(), m0.trainable())):
        diff = (a - b).detach().abs()
        frac, worst = float((diff > 2e-5).float().mean()), float(diff.max())
        assert (frac < 5e-4 and worst < 3e-2) if i < 16 else worst < 1e-4, (i, frac, worst)
    assert loop.losses() == pytest.approx(l1, rel=1e-6)
    # Adam's state sits in the optimiser in torch's own format: step counts, moments
    st = loop.optimizer.state[m1.sigma_net.params]
    assert float(st["step"]) == 5.0 and st["exp_avg"].shape == (3072,)


def test_one_launch_compositing_equals_the_three_launches_bit_for_bit():
    """rm_composite_train_mse (compositing + background, the MSE's gradient, the compositing backward: one wave per ray in one launch; clean_loss keeps the loss
    value and the books beside the MLP backward) against the three launches in a row: six captured steps with perturbed samples from the same state -- every
    parameter, every loss value, the image and the gradients handed to the field's backward, bit for bit."""
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    o, d = _patch_rays(16)
    target = torch.tensor([0.2, 0.5, 0.8]).view(1, 3).expand(256, 3).contiguous().cuda()
    data = {"rays_o": o.cuda()[None], "rays_d": d.cuda()[None], "images": target[None]}
    got = []
    for one_launch in (True, False):
        m, _, _ = _clean_model()
        torch.manual_seed(11)      # (the first step's march offsets come from torch's generator)
        loop = GraphedCleanLoop(m, _adam(m), KW, n_rays=256, update_extra_interval=0, perturb=True, fused_composite=one_launch, seed=3)
        loop.step(data)
        for _ in range(5):
            loop.step()
        torch.cuda.synchronize()
        n = int(loop.count_ring[5, 0])
        got.append(([p.detach().clone() for p in m.trainable()], loop.losses(), loop.image_out.clone(), loop.depth_out.clone(), loop.g_sig[:n].clone(),
                    loop.g_rgb[:n].clone(), loop.noises.clone(), int(loop.step_dev), loop.count_ring.clone()))
        loop.close()
    a, b = got
    assert a[1] == b[1] and len(a[1]) == 6 and a[7] == b[7] == 6
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    for k in (2, 3, 4, 5, 6, 8):
        assert torch.equal(a[k], b[k]), k
    assert float(a[4].abs().max()) > 0


def test_table_adam_inside_the_scatter_owners_equals_the_separate_pass_bit_for_bit():
    """hg_levels_scatter_adam (an owner applies torch.optim.Adam's update to the rows it has just summed: no table gradient in memory, no table pass behind the scatter)
    against hg_levels_scatter + opt_adam_dense: six captured steps with perturbed samples from the same state -- every parameter, both moments and the step count of every
    tensor, and every loss value, bit for bit."""
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    o, d = _patch_rays(16)
    target = torch.tensor([0.2, 0.5, 0.8]).view(1, 3).expand(256, 3).contiguous().cuda()
    data = {"rays_o": o.cuda()[None], "rays_d": d.cuda()[None], "images": target[None]}
    got = []
    for fused in (True, False):
        m, _, _ = _clean_model()
        torch.manual_seed(11)
        loop = GraphedCleanLoop(m, _adam(m), KW, n_rays=256, update_extra_interval=0, perturb=True, fused_table_adam=fused, seed=3)
        assert loop.fused_table_adam is fused
        loop.step(data)
        for _ in range(5):
            loop.step()
        torch.cuda.synchronize()
        state = [(float(loop.optimizer.state[p]["step"]), loop.optimizer.state[p]["exp_avg"].clone(), loop.optimizer.state[p]["exp_avg_sq"].clone()) for p in m.trainable()]
        got.append(([p.detach().clone() for p in m.trainable()], loop.losses(), state))
        loop.close()
    a, b = got
    assert a[1] == b[1] and len(a[1]) == 6
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert torch.equal(x, y), i
    for i, (x, y) in enumerate(zip(a[2], b[2])):
        assert x[0] == y[0] == 6.0 and torch.equal(x[1], y[1]) and torch.equal(x[2], y[2]), i
    assert float((a[0][15] - _clean_model()[0].trainable()[15].detach()).abs().max()) > 0      # (the finest table did move)


@pytest.mark.parametrize("bound,trace_dtype", [(1.0, "f32"), (2.0, "f32"), (1.0, "f16")], ids=["one_cascade", "two_cascades", "one_cascade_f16_traces"])
def test_captured_loop_tracks_the_cpu_oracle_over_200_steps(bound, trace_dtype):
    """From a common state, 200 steps of the captured loop and of the CPU oracle (the reference's operator sequence in torch autograd + torch's
    Adam): the loss every 20th step, and the PSNR of the trained render against the target within 0.1 dB."""
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    o, d = _patch_rays(8)
    N = o.shape[0]
    rng = np.random.RandomState(5)
    target = torch.from_numpy((0.25 + 0.5 * rng.rand(N, 3)).astype(np.float32))
    m, bitfield, C = _clean_model(bound=bound, mlp_scale=0.5)
    assert C == (1 if bound == 1.0 else 2)            # bound = 2: the march walks both cascades (raymarching.cu:280-300), the samples' positions span [-2, 2]
    P, S = _oracle_params(m), _scene(m, bitfield)
    leaves = P["base_tables"] + [P["sigma_params"], P["color_params"]]
    opt_cpu = torch.optim.Adam(leaves, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sched = lambda it: 0.1 ** min(it / 200, 1)      # main_nerf.py:127 over this run's length
    loop = GraphedCleanLoop(m, _adam(m), KW, n_rays=N, update_extra_interval=0, perturb=False, lr_lambda=sched, trace_dtype=trace_dtype)
    data = {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": target.cuda()}
    cpu, gpu = [], []
    for it in range(200):
        for g in opt_cpu.param_groups:
            g["lr"] = 1e-2 * sched(it)
        opt_cpu.zero_grad(set_to_none=True)
        out = fr.run_cuda_train(o, d, None, P, S, **{"dt_gamma": 0.0, "max_steps": 1024})
        loss = ((out["image"] - target) ** 2).mean(-1).mean()
        loss.backward()
        opt_cpu.step()
        cpu.append(float(loss.detach()))
        loop.step(data if it == 0 else None)
    gpu = loop.losses()
    assert len(gpu) == 200 and not loop.overflowed()
    every = list(range(0, 200, 20)) + [199]
    print("\nstep: oracle / captured loss  " + "  ".join(f"{i}: {cpu[i]:.3e} / {gpu[i]:.3e}" for i in every))
    np.testing.assert_allclose([gpu[i] for i in every], [cpu[i] for i in every], rtol=2e-2)
    assert cpu[-1] < 0.05 * cpu[0]
    with torch.no_grad():
        img_cpu = fr.run_cuda_train(o, d, None, P, S, dt_gamma=0.0, max_steps=1024)["image"]
        img_gpu = m.render(o.cuda()[None], d.cuda()[None], None, staged=False, bg_color=1, perturb=False, force_all_rays=True, **KW)["image"][0].cpu()
    psnr = lambda x: -10 * np.log10(float(((x - target) ** 2).mean()))
    print(f"PSNR against the target after 200 steps: oracle {psnr(img_cpu):.3f} dB, captured loop {psnr(img_gpu):.3f} dB")
    assert abs(psnr(img_cpu) - psnr(img_gpu)) < 0.1


def test_grid_refresh_runs_in_its_cadence_between_replays():
    """update_extra_state every 16 steps (utils.py:852-857), from the loop's own ring of sample totals; the replayed march reads the re-packed bitfield."""
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    from nerf_signature_amd import raymarching
    o, d = _patch_rays(16)
    target = torch.full((256, 3), 0.5).cuda()
    m, _, _ = _clean_model()
    loop = GraphedCleanLoop(m, _adam(m, 1e-3), KW, n_rays=256, update_extra_interval=16, perturb=True, headroom=3.0)
    data = {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": target}
    counts = []
    for it in range(40):
        loop.step(data if it == 0 else None)
        counts.append(int(loop.count_ring[it % 16, 0]))
        if it in (0, 16, 32):
            assert m.iter_density == it // 16 + 1
    assert m.iter_density == 3 and loop.global_step == 40 and not loop.overflowed()
    assert m.mean_count == int(sum(counts[16:32]) / 16)                           # the refresh at step 32 saw the totals of steps 16..31
    thresh = min(m.mean_density, m.density_thresh)
    assert torch.equal(m.density_bitfield, raymarching.packbits(m.density_grid, thresh))
    # the captured march walks the CURRENT grid: an eager march of the same rays through the same bitfield counts the same samples
    nears, fars = raymarching.near_far_from_aabb(o.cuda(), d.cuda(), m.aabb_train, m.min_near)
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    raymarching.march_rays_train(o.cuda(), d.cuda(), m.bound, m.density_bitfield, m.cascade, m.grid_size, nears, fars, counter, -1, False, 128, True, 0, 1024)
    assert abs(int(counter[0]) - counts[-1]) <= 256                              # (perturbed starts move a ray's count by at most one sample)
    assert all(np.isfinite(loop.losses()))
    m.density_bitfield.zero_()                                                    # ... an emptied grid (in place): the next replay marches nothing
    loop.update_extra_interval = 0
    loss = float(loop.step())
    assert int(loop.count_ring[40 % 16, 0]) == 0 and loss == pytest.approx(0.25, rel=1e-5)      # white background against the 0.5 target


class _TwoRanks:
    """torch.distributed stand-in over two threads of this process (as tests/test_gpu_dp.py): all_reduce only."""

    def __init__(self):
        self.local = threading.local()
        self.barrier = threading.Barrier(2)
        self.turn = threading.Lock()
        self.slots = [None, None]
        self.bytes, self.calls = [0, 0], [0, 0]      # per rank, over all its calls
        self.log = [[], []]                          # per rank: (operation, elements) of every call, in order

    def is_initialized(self):
        return True

    def get_world_size(self):
        return 2

    def get_rank(self):
        return self.local.rank

    def get_backend(self):
        return "nccl"

    def all_reduce(self, t, op=None, async_op=False):
        self.slots[self.local.rank] = t.clone()
        self.turn.release()
        try:
            self.barrier.wait()
            parts = list(self.slots)
            self.barrier.wait()
        finally:
            self.turn.acquire()
        t.copy_(torch.maximum(parts[0], parts[1]) if op == torch.distributed.ReduceOp.MAX else parts[0] + parts[1])
        self.log[self.local.rank].append((str(op).split(".")[-1], t.numel()))
        self.bytes[self.local.rank] += t.numel() * 4
        self.calls[self.local.rank] += 1


def test_two_rank_exchange_equals_the_single_process_gradient(monkeypatch):
    """Two ranks with their own rays, the flat gradient buffer all-reduced (SUM of gradients seeded with 1/world; tables and MLP gradients as two collectives in this eager form):
    every rank then holds the gradient of the single-process step on the concatenated batch, and both take the same optimiser step."""
    import torch.distributed as dist
    from nerf_signature_amd import dp
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    rays = [_patch_rays(8, lo=180), _patch_rays(8, lo=200)]
    rng = np.random.RandomState(3)
    targets = [torch.from_numpy(rng.rand(64, 3).astype(np.float32)) for _ in range(2)]

    def run(o, d, gt, steps=1, sparse=True):
        m, _, _ = _clean_model()
        # (fused_table_adam=False: the single-process reference has to leave its table gradients in memory like the ranks, which exchange them)
        loop = GraphedCleanLoop(m, _adam(m), KW, n_rays=o.shape[0], update_extra_interval=0, perturb=False, capture=False, fused_table_adam=False, sparse_exchange=sparse)
        data = {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": gt.cuda()}
        for _ in range(steps):
            loop.step(data)
        torch.cuda.synchronize()
        return loop, m

    ref, m_ref = run(torch.cat([rays[0][0], rays[1][0]]), torch.cat([rays[0][1], rays[1][1]]), torch.cat(targets))
    group = _TwoRanks()
    for name in ("is_initialized", "get_world_size", "get_rank", "get_backend", "all_reduce"):
        monkeypatch.setattr(dist, name, getattr(group, name))
    results, errors = [None, None], []

    def two_ranks(sparse):
        def rank_main(r):
            group.turn.acquire()
            try:
                group.local.rank = r
                torch.cuda.set_device(0)
                assert dp.exchange_active() and dp.world_size() == 2
                results[r] = run(rays[r][0], rays[r][1], targets[r], sparse=sparse)
            except BaseException as e:      # noqa: BLE001
                errors.append(e)
                group.barrier.abort()
            finally:
                if group.turn.locked():
                    try:
                        group.turn.release()
                    except RuntimeError:
                        pass

        threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        return list(results)

    # the dense exchange first (every table whole: 64 MiB), then the default: the coarse levels as their live rows
    (d0, _), (d1, _) = two_ranks(False)
    dense_bytes = group.bytes[0]
    assert dense_bytes == group.bytes[1] == 3 * d0.flat.numel() * 4 == 3 * (16 * (1 << 19) * 2 + 3072 + 7168) * 4 and group.calls == [6, 6]
    group.bytes, group.calls = [0, 0], [0, 0]
    (l0, m0), (l1, m1) = two_ranks(True)
    # (prepare() runs two warm-up steps in front of the one that counts: three steps' worth; eagerly the buffer travels in two pieces -- tables, then MLP gradients)
    from nerf_signature_amd.stage1 import SPARSE_EXCHANGE_LEVELS, live_rows
    n_live = sum(live_rows(l).numel() for l in SPARSE_EXCHANGE_LEVELS)
    assert SPARSE_EXCHANGE_LEVELS == (0, 1, 2, 3, 4) and n_live == 305480
    assert l0.bytes_exchanged_per_step == (11 * (1 << 19) * 2 + 2 * n_live + 3072 + 7168) * 4 < 0.73 * d0.bytes_exchanged_per_step
    assert group.bytes[0] == group.bytes[1] == 3 * l0.bytes_exchanged_per_step and group.calls == [6, 6]
    print(f"\nstage-1 exchange: {d0.bytes_exchanged_per_step / 2 ** 20:.2f} MiB dense -> {l0.bytes_exchanged_per_step / 2 ** 20:.2f} MiB with levels 0..4 as their {n_live} live rows")
    # the same sums bit for bit: every element has the same two operands either way, and a row outside the live set is an exact zero on every rank
    assert torch.equal(l0.g_tables, d0.g_tables) and torch.equal(l0.g_sigma, d0.g_sigma) and torch.equal(l0.g_color, d0.g_color)
    for l in SPARSE_EXCHANGE_LEVELS:
        dead = torch.ones(1 << 19, dtype=torch.bool, device="cuda")
        dead[live_rows(l).cuda()] = False
        assert float(l0.g_tables[l][dead].abs().max()) == 0.0 and int((l0.g_tables[l] != 0).any(-1).sum()) > 0, l
    assert torch.equal(l0.flat, l1.flat)
    print(f"\ntwo-rank vs single-process stage-1 gradient: rel. L2 tables {rel(l0.g_tables, ref.g_tables):.2e}, sigma MLP {rel(l0.g_sigma, ref.g_sigma):.2e}, "
          f"colour MLP {rel(l0.g_color, ref.g_color):.2e}; loss {0.5 * (float(l0.loss) + float(l1.loss)):.6f} vs {float(ref.loss):.6f}")
    assert rel(l0.g_tables, ref.g_tables) < 1e-5 and rel(l0.g_sigma, ref.g_sigma) < 1e-5 and rel(l0.g_color, ref.g_color) < 1e-5
    assert 0.5 * (float(l0.loss) + float(l1.loss)) == pytest.approx(float(ref.loss), rel=1e-5)
    for a, b in zip(m0.trainable(), m1.trainable()):
        assert torch.equal(a, b)                                                  # replicas in lockstep


def test_one_rank_nearing_its_capacity_makes_both_ranks_grow_in_the_same_step(monkeypatch):
    """ADVICE round 5 (stage1.py:542): the decision to grow the point buffers and capture again is taken from the maximum of the window's sample totals over ALL
    ranks (read without synchronising, one refresh late: the refresh at step 4 queues the window's peak, the one at step 8 acts on it).  Rank 0's rays fill 95 % of
    the buffers, rank 1's a fraction: both grow at the same refresh to the same capacity, issue the same sequence of collectives
    (a rank growing alone would run its warm-up steps' collectives against its peer's ordinary step) and stay in lockstep."""
    import torch.distributed as dist
    from nerf_signature_amd import dp
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    rays = [_patch_rays(16, lo=184), _patch_rays(4, lo=20)]                           # through the middle of the ball | 16 rays near the image corner
    rays[1] = tuple(t.repeat(16, 1) for t in rays[1])                                # (same number of rays on both ranks)
    target = torch.full((256, 3), 0.5)

    def run(r, capacity, steps):
        m, _, _ = _clean_model()
        loop = GraphedCleanLoop(m, _adam(m, 1e-3), KW, n_rays=256, update_extra_interval=4, perturb=False, capture=False, fused_table_adam=False, capacity=capacity)
        data = {"rays_o": rays[r][0].cuda(), "rays_d": rays[r][1].cuda(), "images": target.cuda()}
        totals = []
        for it in range(steps):
            loop.step(data if it == 0 else None)
            totals.append(int(loop.count_ring[it % 16, 0]))
        torch.cuda.synchronize()
        return loop, m, totals

    from nerf_signature_amd.raymarching import padded_point_count
    n0, n1 = (run(r, None, 1)[2][0] for r in range(2))
    capacity = padded_point_count(int(n0 / 0.95))
    assert n1 < 0.5 * capacity < 0.9 * capacity < n0 <= capacity, (n0, n1, capacity)
    group = _TwoRanks()
    for name in ("is_initialized", "get_world_size", "get_rank", "get_backend", "all_reduce"):
        monkeypatch.setattr(dist, name, getattr(group, name))
    results, errors = [None, None], []

    def rank_main(r):
        group.turn.acquire()
        try:
            group.local.rank = r
            torch.cuda.set_device(0)
            results[r] = run(r, capacity, 10)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)
            group.barrier.abort()
        finally:
            if group.turn.locked():
                try:
                    group.turn.release()
                except RuntimeError:
                    pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    (l0, m0, t0), (l1, m1, t1) = results
    assert l0.recaptures == l1.recaptures == 1 and l0.capacity == l1.capacity > capacity          # rank 1 alone would not have grown
    assert group.log[0] == group.log[1] and ("MAX", 1) in group.log[0]                             # the same collectives, in the same order, on both ranks
    assert l0.global_step == l1.global_step == 10 and not l0.overflowed() and not l1.overflowed()
    for a, b in zip(m0.trainable(), m1.trainable()):
        assert torch.equal(a, b)                                                                   # ... and the replicas are still in lockstep
    for p in m0.trainable():
        assert float(l0.optimizer.state[p]["step"]) == 10.0                                        # ten optimiser steps: the warm-up of the re-capture trained nothing


@pytest.mark.parametrize("bound", [1.0, 2.0], ids=["one_cascade", "two_cascades"])
def test_grid_refresh_full_and_partial_over_the_cascades_captured_equals_eager(bound):
    """update_extra_state in BOTH of its forms (renderer_wtmk.py:462-514: the first 16 refreshes probe every cell of every cascade, later ones a random quarter plus
    as many occupied cells per cascade) inside the stage-1 loop: 40 steps with a refresh every 2 steps = 16 full + 4 partial refreshes.  The captured loop and the
    same kernel sequence issued eagerly, from one seed, leave the same density grid, bitfield, mean density, sample totals and parameters bit for bit; on two
    cascades both cascades' grids move."""
    from nerf_signature_amd.stage1 import GraphedCleanLoop
    from nerf_signature_amd import raymarching
    o, d = _patch_rays(16)
    target = torch.full((256, 3), 0.5).cuda()

    def run(capture):
        torch.manual_seed(7)                                                         # the partial refresh draws its cells from torch's generator
        m, _, C = _clean_model(bound=bound)
        grid0 = m.density_grid.clone()
        loop = GraphedCleanLoop(m, _adam(m, 1e-3), KW, n_rays=256, update_extra_interval=2, perturb=False, headroom=3.0, capture=capture)
        data = {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": target}
        totals = []
        for it in range(40):
            loop.step(data if it == 0 else None)
            totals.append(int(loop.count_ring[it % 16, 0]))
        torch.cuda.synchronize()
        assert m.iter_density == 20 and loop.global_step == 40 and not loop.overflowed() and loop.recaptures == 0
        return m, loop, totals, grid0, C

    m_c, loop_c, totals_c, grid0, C = run(True)
    m_e, loop_e, totals_e, _, _ = run(False)
    assert C == (1 if bound == 1.0 else 2)
    assert totals_c == totals_e and sum(totals_c) > 0
    assert torch.equal(m_c.density_grid, m_e.density_grid) and torch.equal(m_c.density_bitfield, m_e.density_bitfield)
    assert m_c.mean_density == m_e.mean_density and m_c.mean_count == m_e.mean_count == int(sum(totals_c[36:38]) / 2)      # the refresh at step 38 saw steps 36, 37
    for a, b in zip(m_c.trainable(), m_e.trainable()):
        assert torch.equal(a, b)
    for c in range(C):
        assert not torch.equal(m_c.density_grid[c], grid0[c].cuda()), c              # every cascade's grid was refreshed
    thresh = min(m_c.mean_density, m_c.density_thresh)
    assert torch.equal(m_c.density_bitfield, raymarching.packbits(m_c.density_grid, thresh))
    assert all(np.isfinite(loop_c.losses()))


def test_parameter_ema_follows_torch_ema_inside_the_captured_step():
    """ema_decay (the stage-1 trainer's ExponentialMovingAverage(model.parameters(), decay=0.95), main_nerf.py:130, utils.py:389-390,761-762): the shadow the
    loop keeps inside its captured step equals torch_ema 0.3's update() restated with tensor operators on the parameters after every step -- its warm-up
    decay min(0.95, (1 + n) / (10 + n)) included -- bit for bit; the eager autograd loop's restatement agrees with its own parameters the same way; the set-up's
    warm-up steps leave the shadow alone; ema_weights() evaluates with the averages and puts the trained parameters back."""
    from nerf_signature_amd.stage1 import CleanLoop, GraphedCleanLoop
    o, d = _patch_rays(8)
    rng = np.random.RandomState(11)
    target = torch.from_numpy((0.25 + 0.5 * rng.rand(64, 3)).astype(np.float32)).cuda()
    data = {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": target}

    def torch_ema(shadow, params, n):
        decay = min(0.95, (1 + n) / (10 + n))
        for s_, p in zip(shadow, params):
            tmp = s_ - p
            tmp.mul_(1.0 - decay)
            s_.sub_(tmp)

    finals = []
    for capture in (False, True):
        m, _, _ = _clean_model(mlp_scale=0.5)
        loop = GraphedCleanLoop(m, _adam(m), KW, n_rays=64, update_extra_interval=0, perturb=False, capture=capture, ema_decay=0.95)
        start = [p.detach().clone() for p in m.trainable()]
        want = [p.clone() for p in start]
        for k in range(1, 8):
            loop.step(data if k == 1 else None)
            if k == 1:
                pass
            torch_ema(want, [p.detach() for p in m.trainable()], k)
            if not capture or k == 7:
                for a, b in zip(loop.ema_parameters(), want):
                    assert torch.equal(a, b), (capture, k)
        assert not torch.equal(loop.ema_parameters()[16], m.trainable()[16]) and not torch.equal(loop.ema_parameters()[16], start[16])
        trained = [p.detach().clone() for p in m.trainable()]
        with loop.ema_weights():
            for p, s_ in zip(m.trainable(), loop.ema_parameters()):
                assert torch.equal(p, s_)
            img = m.render(o.cuda()[None], d.cuda()[None], None, staged=False, bg_color=1, perturb=False, force_all_rays=True, **KW)["image"]
            assert bool(torch.isfinite(img).all())
        for p, t in zip(m.trainable(), trained):
            assert torch.equal(p, t)
        finals.append([t.clone() for t in loop.ema_parameters()])
    for a, b in zip(*finals):
        assert torch.equal(a, b)                                                  # captured == eager
    # the autograd loop's restatement (CleanLoop): the same recurrence on its own parameters
    m, _, _ = _clean_model(mlp_scale=0.5)
    eager = CleanLoop(m, _adam(m), KW, update_extra_interval=10 ** 9, ema_decay=0.95)
    eager.global_step = 1
    want = [p.detach().clone() for p in m.trainable()]
    batch = {"rays_o": o.cuda()[None], "rays_d": d.cuda()[None], "images": target[None], "perturb": False, "force_all_rays": True}
    for k in range(1, 4):
        eager.step(batch)
        torch_ema(want, [p.detach() for p in m.trainable()], k)
    for a, b in zip(eager.ema_shadow, want):
        assert torch.equal(a, b)
