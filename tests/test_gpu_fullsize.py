"""The bench workload itself (BASELINE.json configs[1]: hotdog-like scene S0, 4608 block rays + 4096 content rays, 32-bit
message, 1.4 M sample points) through the C ABI at FULL size.  Integer outputs are compared with the C oracle outright (the
scalar march of 8704 rays takes a second); the floating-point stages, too large for the CPU oracle, are checked through
size-independent properties plus a random sample of points against the oracle."""
import numpy as np
import pytest
import torch

from oracle import field_ref as fr
from oracle import raymarch_ref as orm

pytestmark = pytest.mark.gpu
D = 32


@pytest.fixture(scope="module")
def workload():
    from nerf_signature_amd import synthetic
    from nerf_signature_amd.network import NeRFNetwork
    torch.manual_seed(0)
    m = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
    synthetic.init_model(m, "hotdog")
    m.cuda().train()
    bo, bd = synthetic.block_rays("hotdog")
    co, cd = synthetic.content_rays("hotdog", 4096, seed=0)
    o = torch.cat([bo.reshape(-1, 3), co.reshape(-1, 3)]).contiguous()
    d = torch.cat([bd.reshape(-1, 3), cd.reshape(-1, 3)]).contiguous()
    return m, o, d


@pytest.fixture(scope="module")
def marched(workload):
    from nerf_signature_amd import raymarching as rm
    m, o, d = workload
    oc, dc = o.cuda(), d.cuda()
    nears, fars = rm.near_far_from_aabb(oc, dc, m.aabb_train, 0.2)
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
    xyzs, dirs, deltas, rays = rm.march_rays_train(oc, dc, 1.0, m.density_bitfield, 1, 128, nears, fars, ctr, -1, False, 128, True, 0.0, 1024)
    return nears, fars, ctr, xyzs, dirs, deltas, rays


def test_full_workload_march_is_bit_exact(workload, marched):
    """Every ray record (id, offset, count), every sample position / direction / delta of the 8704-ray workload == the C oracle."""
    m, o, d = workload
    nears, fars, ctr, xyzs, dirs, deltas, rays = marched
    bitfield = m.density_bitfield.cpu().numpy()
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    n0, f0 = orm.near_far_from_aabb(o.numpy(), d.numpy(), aabb, 0.2)
    np.testing.assert_array_equal(nears.cpu().numpy(), n0)
    np.testing.assert_array_equal(fars.cpu().numpy(), f0)
    ctr0 = np.zeros(2, np.int32)
    x0, d0, dl0, rays0 = orm.march_rays_train(o.numpy(), d.numpy(), 1.0, bitfield, 1, 128, n0, f0, ctr0, -1, False, 128, True, 0.0, 1024)
    assert ctr0[0] > 1_400_000 and ctr0[1] == 8704
    np.testing.assert_array_equal(ctr.cpu().numpy(), ctr0)
    np.testing.assert_array_equal(rays.cpu().numpy(), rays0)
    np.testing.assert_array_equal(xyzs.cpu().numpy(), x0)
    np.testing.assert_array_equal(dirs.cpu().numpy(), d0)
    np.testing.assert_array_equal(deltas.cpu().numpy(), dl0)
    # structure: offsets are the exclusive prefix sum of the counts in ray-id order; the padded tail rows are zero
    r = rays.cpu().numpy().astype(np.int64)
    assert np.array_equal(r[:, 0], np.arange(8704)) and np.array_equal(r[:, 1], np.concatenate([[0], np.cumsum(r[:, 2])[:-1]]))
    M = int(ctr0[0])
    assert xyzs.shape[0] == M + (128 - M % 128) and float(xyzs[M:].abs().max()) == 0.0


def test_full_workload_encoder_linearity_and_sampled_parity(workload, marched):
    """1.4 M points: (a) the pre-summed codebook gather equals the literal sum of D gathers (linearity of the interpolation in the
    table); (b) the plane-layout encoder used by the training path is bit-identical to the row-layout one;
    (c) 4096 randomly drawn points agree with the oracle bit for bit (base features) / to 1e-6 (codebook sum)."""
    from nerf_signature_amd import fieldops as fo
    m, _, _ = workload
    xyzs = marched[3]
    M = int(marched[2][0])
    x01 = ((xyzs[:M] + 1.0) / 2.0).contiguous()
    msg = torch.from_numpy(np.random.RandomState(3).randint(0, 2, D).astype(np.float32))
    base = m.encoder.tables()
    sel = fo.select_tables(m.msg_encoder.tables(), fo.message_bits(msg))
    S = fo.codebook_presum(sel)
    feat = fo.encode(x01, base, S)                                         # [M,32], codebook already added into 30:32
    feat_base = fo.encode(x01, base, None)
    literal = fo.codebook_encode_literal(x01, sel)                         # D separate gathers, summed
    np.testing.assert_allclose((feat[:, 30:] - feat_base[:, 30:]).cpu().numpy(), literal.cpu().numpy(), rtol=0, atol=2e-6)
    assert torch.equal(feat[:, :30], feat_base[:, :30])
    pick = torch.from_numpy(np.random.RandomState(4).choice(M, 4096, replace=False)).cuda()
    xs = x01[pick].cpu()
    ref_base = fr.base_encode(xs, [t.detach().cpu() for t in base])
    np.testing.assert_array_equal(feat_base[pick].cpu().numpy(), ref_base.numpy())
    ref_cb = fr.codebook_encode(xs, msg, [t.detach().cpu() for t in m.msg_encoder.tables()])
    np.testing.assert_allclose(literal[pick].cpu().numpy(), ref_cb.numpy(), rtol=0, atol=1e-6)
    # (b) the training forward (planes route for large batches) against the fused-gather route on all rows
    packed = m._packed()
    dirs = marched[4]
    a = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=True)
    b = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])


def test_full_workload_field_and_composite_properties(workload, marched):
    """sigma / rgb of a 4096-point sample against the oracle's fp32 network (1e-3, north_star); compositing of all 8704 rays against
    a cumprod formulation on the device (renderer_wtmk.py:205-229 style); gradient of the composite w.r.t. sigma / rgb against
    autograd of that formulation."""
    from nerf_signature_amd import fieldops as fo
    from nerf_signature_amd import raymarching as rm
    m, _, _ = workload
    nears, fars, ctr, xyzs, dirs, deltas, rays = marched
    M = int(ctr[0])
    msg = torch.from_numpy(np.random.RandomState(3).randint(0, 2, D).astype(np.float32))
    base = m.encoder.tables()
    sel = fo.select_tables(m.msg_encoder.tables(), fo.message_bits(msg))
    S = fo.codebook_presum(sel)
    sig, rgb = fo.field_forward(xyzs, dirs, 1.0, base, S, m._packed())[:2]
    pick = torch.from_numpy(np.random.RandomState(5).choice(M, 4096, replace=False)).cuda()
    P = {"bound": 1.0, "base_tables": [t.detach().cpu() for t in base], "cb_tables": [t.detach().cpu() for t in m.msg_encoder.tables()],
         "sigma_params": m.sigma_net.params.detach().cpu(), "color_params": m.color_net.params.detach().cpu()}
    with torch.no_grad():
        s0, c0 = fr.field_forward(xyzs[pick].cpu(), dirs[pick].cpu(), msg, P)
    np.testing.assert_allclose(sig[pick].cpu().numpy(), s0.numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(rgb[pick].cpu().numpy(), c0.numpy(), rtol=0, atol=1e-3)

    sig_g, rgb_g = sig.detach().clone().requires_grad_(True), rgb.detach().float().clone().requires_grad_(True)
    ws, depth, image = rm.composite_rays_train(sig_g, rgb_g, deltas, rays, 1e-4)
    # reference formulation, ray by ray segments on the device: alpha, transmittance by exclusive cumprod, early stop at T < 1e-4
    r = rays.long()
    N, off, cnt = r.shape[0], r[:, 1], r[:, 2]
    L = int(cnt.max())
    k = torch.arange(L, device="cuda")[None, :]
    valid = k < cnt[:, None]
    idx = (off[:, None] + k).clamp(max=xyzs.shape[0] - 1)
    sg = torch.where(valid, sig.detach()[idx], torch.zeros((), device="cuda")).double().requires_grad_(True)
    cg = torch.where(valid[..., None], rgb.detach().float()[idx], torch.zeros((), device="cuda")).double().requires_grad_(True)
    alpha = 1.0 - torch.exp(-sg * deltas[idx][..., 0].double())
    T = torch.cumprod(torch.cat([torch.ones(N, 1, device="cuda", dtype=torch.float64), 1.0 - alpha], dim=1), dim=1)[:, :-1]
    live = valid & (T >= 1e-4)                 # the kernel stops a ray once T drops below the threshold (raymarching.cu:543)
    w = torch.where(live, alpha * T, torch.zeros((), device="cuda", dtype=torch.float64))
    ws0, img0 = w.sum(1), (w[..., None] * cg).sum(1)
    np.testing.assert_allclose(ws.detach().cpu().numpy(), ws0.detach().cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(image.detach().cpu().numpy(), img0.detach().cpu().numpy(), rtol=0, atol=2e-4)
    assert float(ws.detach().min()) >= 0.0 and float(ws.detach().max()) <= 1.0 + 1e-5
    gw, gi = torch.randn(N, device="cuda"), torch.randn(N, 3, device="cuda")
    (ws * gw).sum().add((image * gi).sum()).backward()
    (ws0 * gw.double()).sum().add((img0 * gi.double()).sum()).backward()
    g_sig0 = torch.zeros_like(sig_g).double().index_put_((idx[valid],), sg.grad[valid], accumulate=True)
    g_rgb0 = torch.zeros_like(rgb_g).double().index_put_((idx[valid],), cg.grad[valid], accumulate=True)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    assert rel(sig_g.grad, g_sig0) < 1e-3 and rel(rgb_g.grad, g_rgb0) < 1e-3


def test_full_workload_block_render_with_kept_planes_is_bit_identical(workload):
    """The bench's 4608 block rays (1.29 M points) declared constant (NeRFNetwork.fix_rays): image, depth and weights of the render that
    gathers only the codebook level per step == the render that recomputes all 17 levels, bit for bit, for two messages; the
    codebook gradient agrees to the order of the sums (both routes are the fixed-point slice-binned scatter at this size)."""
    from nerf_signature_amd import synthetic
    m, _, _ = workload
    bo, bd = (t.cuda() for t in synthetic.block_rays("hotdog"))
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
    gvec = torch.rand(bo.shape, device="cuda")
    msgs = [torch.from_numpy(np.random.RandomState(s).randint(0, 2, D).astype(np.float32)) for s in (1, 2)]

    def render(o, d, msg):
        for e in m.msg_encoder.embeddings:
            e.weight.grad = None
        out = m.render(o, d, msg, **kw)
        (out["image"] * gvec).sum().backward()
        grad = m.msg_encoder.embeddings[int(msg[0])].weight.grad
        return out["image"].detach().clone(), out["weights_sum"].detach().clone(), out["depth"].detach().clone(), grad.clone()

    want = [render(bo.clone(), bd.clone(), msg) for msg in msgs]
    rec = m.fix_rays(bo, bd, dt_gamma=0, max_steps=1024)
    try:
        assert int(rec["counter"][0]) > 1_250_000
        for msg, w in zip(msgs, want):
            got = render(bo, bd, msg)
            assert torch.equal(got[0], w[0]) and torch.equal(got[1], w[1]) and torch.allclose(got[2], w[2], rtol=0, atol=0, equal_nan=True)
            assert float((got[3] - w[3]).norm() / w[3].norm()) < 1e-6
        assert rec["fixed"].refreshes == 1
    finally:
        m.drop_marched()
