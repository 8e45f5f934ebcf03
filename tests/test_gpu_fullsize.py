"""The bench workloads themselves through the C ABI at FULL size:
  * BASELINE.json configs[1] (hotdog-like scene S0: 4608 block rays + 4096 content rays, 32-bit message, 1.4 M sample points),
  * configs[2] (Mip-NeRF360/counter-like scene S1: bound 2, TWO cascades, camera inside the box, the same 4608 + 4096 rays at
    ~500 / ~240 samples per ray = 3 M points -- capacity sizing, cascade selection `mip_from_pos` / `mip_from_dt`
    (raymarching.cu:42-54,368) and the cascade-1 dt_max at the size the secondary bench line times),
  * configs[4] (LLFF/fern-like scene S2: 1008x756 view staged in 187 chunks of 4096 rays, 48-bit message, 64x64 block grid,
    48 blocks of 11x15 through the HiDDeN decoder).
Integer outputs are compared with the C oracle outright (the scalar march of 8704 rays takes seconds); the floating-point
stages, too large for the CPU oracle, are checked through size-independent properties plus a random sample of points /
pixels against the oracle."""
import numpy as np
import pytest
import torch

from oracle import field_ref as fr
from oracle import raymarch_ref as orm

pytestmark = pytest.mark.gpu
D = 32


def _build(scene, message_dim):
    from nerf_signature_amd import synthetic
    from nerf_signature_amd.network import NeRFNetwork
    torch.manual_seed(0)
    cfg = synthetic.SCENES[scene]
    m = NeRFNetwork(bound=cfg["bound"], cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=message_dim, n_views=1)
    synthetic.init_model(m, scene)
    return m.cuda().train(), cfg


@pytest.fixture(scope="module", params=["hotdog", "counter"])
def workload(request):
    """(model, rays_o, rays_d) of one training step of the scene: its 4608 block rays followed by 4096 content rays (CPU tensors)."""
    from nerf_signature_amd import synthetic
    scene = request.param
    m, cfg = _build(scene, D)
    bo, bd = synthetic.block_rays(scene)
    co, cd = synthetic.content_rays(scene, 4096, seed=0)
    o = torch.cat([bo.reshape(-1, 3), co.reshape(-1, 3)]).contiguous()
    d = torch.cat([bd.reshape(-1, 3), cd.reshape(-1, 3)]).contiguous()
    m.scene_name, m.scene_cfg = scene, cfg
    yield m, o, d
    del m
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def marched(workload):
    from nerf_signature_amd import raymarching as rm
    m, o, d = workload
    oc, dc = o.cuda(), d.cuda()
    nears, fars = rm.near_far_from_aabb(oc, dc, m.aabb_train, 0.2)
    ctr = torch.zeros(2, dtype=torch.int32, device="cuda")
    xyzs, dirs, deltas, rays = rm.march_rays_train(oc, dc, m.bound, m.density_bitfield, m.cascade, 128, nears, fars, ctr, -1, False, 128, True,
                                                   m.scene_cfg["dt_gamma"], 1024)
    return nears, fars, ctr, xyzs, dirs, deltas, rays


def test_full_workload_march_is_bit_exact(workload, marched):
    """Every ray record (id, offset, count), every sample position / direction / delta of the 8704-ray workload == the C oracle."""
    m, o, d = workload
    nears, fars, ctr, xyzs, dirs, deltas, rays = marched
    bitfield = m.density_bitfield.cpu().numpy()
    b = float(m.bound)
    aabb = np.array([-b, -b, -b, b, b, b], np.float32)
    n0, f0 = orm.near_far_from_aabb(o.numpy(), d.numpy(), aabb, 0.2)
    np.testing.assert_array_equal(nears.cpu().numpy(), n0)
    np.testing.assert_array_equal(fars.cpu().numpy(), f0)
    ctr0 = np.zeros(2, np.int32)
    x0, d0, dl0, rays0 = orm.march_rays_train(o.numpy(), d.numpy(), b, bitfield, m.cascade, 128, n0, f0, ctr0, -1, False, 128, True,
                                              m.scene_cfg["dt_gamma"], 1024)
    # hotdog: 280 / 30 samples per block / content ray; counter (camera inside, shell on the coarse cascade): ~500 / ~240
    assert ctr0[0] > {"hotdog": 1_400_000, "counter": 2_500_000}[m.scene_name] and ctr0[1] == 8704
    if m.scene_name == "counter":       # both cascades are really sampled: points inside |p| < 1 (fine cells) and in the shell beyond 1.5 (coarse cells)
        linf = np.abs(x0[:int(ctr0[0])]).max(-1)
        assert m.cascade == 2 and int((linf > 1.5).sum()) > 100_000 and int((linf < 1.0).sum()) > 100_000
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
    x01 = ((xyzs[:M] + m.bound) / (2.0 * m.bound)).contiguous()
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
    a = fo.field_forward(xyzs, dirs, m.bound, base, S, packed, want_masks=True, planes=True)
    b = fo.field_forward(xyzs, dirs, m.bound, base, S, packed, want_masks=True, planes=False)
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
    sig, rgb = fo.field_forward(xyzs, dirs, m.bound, base, S, m._packed())[:2]
    pick = torch.from_numpy(np.random.RandomState(5).choice(M, 4096, replace=False)).cuda()
    P = {"bound": float(m.bound), "base_tables": [t.detach().cpu() for t in base], "cb_tables": [t.detach().cpu() for t in m.msg_encoder.tables()],
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
    bo, bd = (t.cuda() for t in synthetic.block_rays(m.scene_name))
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=m.scene_cfg["dt_gamma"], max_steps=1024)
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
    rec = m.fix_rays(bo, bd, dt_gamma=m.scene_cfg["dt_gamma"], max_steps=1024)
    try:
        assert int(rec["counter"][0]) > 1_250_000
        for msg, w in zip(msgs, want):
            got = render(bo, bd, msg)
            assert torch.equal(got[0], w[0]) and torch.equal(got[1], w[1]) and torch.allclose(got[2], w[2], rtol=0, atol=0, equal_nan=True)
            assert float((got[3] - w[3]).norm() / w[3].norm()) < 1e-6
        assert rec["fixed"].refreshes == 1
    finally:
        m.drop_marched()


# ------------------------------------------------------------------------------------------------ config 3 as a whole training step

def test_counter_full_train_step_sampled_rays_vs_oracle():
    """One full-size training-mode render of each kind on scene S1 (4608 block rays, 4096 content rays, two cascades) against the oracle on a
    random sample of 192 rays each (the oracle's result for a ray does not depend on which other rays are in the batch): image, weights,
    depth within 1e-3; the whole batch's point count equals the C oracle's."""
    from nerf_signature_amd import synthetic
    m, cfg = _build("counter", D)
    msg = torch.from_numpy(np.random.RandomState(11).randint(0, 2, D).astype(np.float32))
    P = {"bound": 2.0, "base_tables": [t.detach().cpu() for t in m.encoder.tables()], "cb_tables": [t.detach().cpu() for t in m.msg_encoder.tables()],
         "sigma_params": m.sigma_net.params.detach().cpu(), "color_params": m.color_net.params.detach().cpu()}
    S = {"bound": 2.0, "cascade": 2, "grid_size": 128, "density_bitfield": m.density_bitfield.cpu().numpy(), "aabb": np.array([-2, -2, -2, 2, 2, 2], np.float32),
         "min_near": 0.2, "density_scale": 1}
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
    bo, bd = synthetic.block_rays("counter")
    co, cd = synthetic.content_rays("counter", 4096, seed=0)
    for o, d in ((bo.reshape(1, -1, 3), bd.reshape(1, -1, 3)), (co, cd)):
        with torch.no_grad():
            out = m.render(o.cuda(), d.cuda(), msg, **kw)
        n_points = int(m.step_counter[(m.local_step - 1) % 16, 0])
        nears, fars = orm.near_far_from_aabb(o[0].numpy(), d[0].numpy(), S["aabb"], 0.2)
        _, total = orm.march_counts(o[0].numpy(), d[0].numpy(), 2.0, S["density_bitfield"], 2, 128, nears, fars, 0.0, 1024)
        assert n_points == total > 900_000
        pick = np.random.RandomState(12).choice(o.shape[1], 192, replace=False)
        with torch.no_grad():
            ref = fr.render(o[:, pick], d[:, pick], msg, P, S, staged=False, bg_color=1, dt_gamma=0.0, max_steps=1024)
        np.testing.assert_allclose(out["image"][0, pick].cpu().numpy(), ref["image"][0].numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(out["weights_sum"][pick].cpu().numpy(), ref["weights_sum"].numpy(), rtol=0, atol=1e-3)
        hit = ~torch.isnan(ref["depth"][0])
        np.testing.assert_allclose(out["depth"][0, pick].cpu()[hit].numpy(), ref["depth"][0][hit].numpy(), rtol=0, atol=1e-3)


# ------------------------------------------------------------------------------------------------ config 5: fern, full image

@pytest.fixture(scope="module")
def fern():
    from nerf_signature_amd import rays, synthetic
    m, cfg = _build("fern", 48)
    H, W = cfg["H"], cfg["W"]
    intr = (cfg["focal"], cfg["focal"], W / 2, H / 2)
    pose = torch.from_numpy(synthetic.orbit_pose(1.1, 0.7, cfg["radius"]))[None].cuda()
    r = rays.get_rays(pose, intr, H, W, -1)
    msg = torch.from_numpy(np.random.RandomState(5).randint(0, 2, 48).astype(np.float32))
    P = {"bound": 2.0, "base_tables": [t.detach().cpu() for t in m.encoder.tables()], "cb_tables": [t.detach().cpu() for t in m.msg_encoder.tables()],
         "sigma_params": m.sigma_net.params.detach().cpu(), "color_params": m.color_net.params.detach().cpu()}
    S = {"bound": 2.0, "cascade": 2, "grid_size": 128, "density_bitfield": m.density_bitfield.cpu().numpy(), "aabb": np.array([-2, -2, -2, 2, 2, 2], np.float32),
         "min_near": 0.2, "density_scale": 1}
    yield m, cfg, r["rays_o"], r["rays_d"], msg, P, S
    del m
    torch.cuda.empty_cache()


def test_fern_full_image_fused_staging_equals_chunk_by_chunk_and_sampled_pixels_vs_oracle(fern, monkeypatch):
    """The 1008x756 view (762 048 rays = 187 chunks of 4096, renderer_wtmk.py:555-570) under no_grad with the model in training mode, as
    the reference's test_image runs it: (a) the fused staged render (up to 64 chunks per launch sequence) is BIT-IDENTICAL to walking the
    image chunk by chunk, image and depth, and leaves the same local_step / step_counter ring behind; (b) 2048 random pixels agree with
    the oracle's render of exactly those rays to 1e-3 (image, depth)."""
    m, cfg, ro, rd, msg, P, S = fern
    H, W = cfg["H"], cfg["W"]
    assert ro.shape == (1, H * W, 3) and (H * W + 4095) // 4096 == 187
    kw = dict(staged=True, max_ray_batch=4096, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=cfg["dt_gamma"], max_steps=1024)
    with torch.no_grad():
        m.local_step = 0
        m.step_counter.zero_()
        fused = m.render(ro, rd, msg, **kw)
        ring_fused, steps_fused = m.step_counter.clone(), m.local_step
        monkeypatch.setenv("NERFSIG_STAGED_FUSED", "0")
        m.local_step = 0
        m.step_counter.zero_()
        chunked = m.render(ro, rd, msg, **kw)
        ring_chunked, steps_chunked = m.step_counter.clone(), m.local_step
        monkeypatch.delenv("NERFSIG_STAGED_FUSED")
    assert torch.equal(fused["image"], chunked["image"])
    assert torch.allclose(fused["depth"], chunked["depth"], rtol=0, atol=0, equal_nan=True)
    assert steps_fused == steps_chunked == 187 and torch.equal(ring_fused, ring_chunked)
    assert 0.05 < float((fused["image"] < 0.999).any(-1).float().mean())          # a real picture, not only background
    pick = np.random.RandomState(21).choice(H * W, 2048, replace=False)
    with torch.no_grad():
        ref = fr.render(ro[:, pick].cpu(), rd[:, pick].cpu(), msg, P, S, staged=False, bg_color=1, dt_gamma=cfg["dt_gamma"], max_steps=1024)
    assert ref["n_points"] > 50_000
    np.testing.assert_allclose(fused["image"][0, pick].cpu().numpy(), ref["image"][0].numpy(), rtol=0, atol=1e-3)
    hit = ~torch.isnan(ref["depth"][0])
    np.testing.assert_allclose(fused["depth"][0, pick].cpu()[hit].numpy(), ref["depth"][0][hit].numpy(), rtol=0, atol=1e-3)


def test_fern_48_blocks_decoder_logits_and_bit_accuracy_vs_oracle(fern):
    """The 48 selected 11x15 blocks (64x64 block grid over 1008x756) rendered with the message and decoded (eval_step's block branch,
    utils_wtmk_disen.py:662-680): rendered blocks, decoder logits and bit accuracy against the oracle's render of the same 7920 rays
    through the stock decoder on the CPU."""
    import copy
    from nerf_signature_amd import synthetic
    from nerf_signature_amd.trainer import BIT_ACC
    m, cfg, _, _, msg, P, S = fern
    bo, bd = synthetic.block_rays("fern")
    assert bo.shape == (48, 11, 15, 3)
    with torch.no_grad():
        out = m.render(bo.cuda(), bd.cuda(), msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=cfg["dt_gamma"], max_steps=1024)
        img1 = out["image"].clamp(0, 1)
        dec1 = m.msg_decoder(m.normalization(img1.permute(0, 3, 1, 2)))
        ref = fr.render(bo.reshape(1, -1, 3), bd.reshape(1, -1, 3), msg, P, S, staged=True, max_ray_batch=11 * 15 * 6, bg_color=1,
                        dt_gamma=cfg["dt_gamma"], max_steps=1024)
        img0 = ref["image"].reshape(48, 11, 15, 3).clamp(0, 1)
        dec0 = copy.deepcopy(m.msg_decoder).cpu()(fr.normalize_img(img0.permute(0, 3, 1, 2)))
    np.testing.assert_allclose(img1.cpu().numpy(), img0.numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(dec1.cpu().numpy(), dec0.numpy(), rtol=0, atol=2e-3)
    a1, a0 = BIT_ACC(), BIT_ACC()
    a1.update(dec1.cpu().permute(1, 0), msg[None])
    a0.update(dec0.permute(1, 0), msg[None])
    assert abs(a1.measure() - a0.measure()) <= 1.0 / 48                 # within one bit (north_star)
