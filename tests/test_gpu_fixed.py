"""Rays that do not change between training steps (NeRFNetwork.fix_rays, fieldops.FixedPoints, GraphedWatermarkLoop(fixed_blocks=True)):
the watermark-block rays are one pair of tensors per dataset in the reference (nerf/provider_wtmk.py:442-494) and everything their
field pass reads except the codebook is frozen in the watermark stage (network_wtmk_tcnn.py:90-95), so their samples, their 16
base-level feature planes and their scatter plan are computed once and only the codebook level is gathered per step.  The bar is
identity with the path that recomputes everything: same kernels' arithmetic on the same inputs."""
import numpy as np
import pytest
import torch

import closed_form as cf
from test_gpu_render import _data, _model

pytestmark = pytest.mark.gpu
KW = dict(staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)


def test_fixed_rays_render_is_bit_identical_and_follows_its_inputs():
    m, _, _ = _model()
    bo, bd, _, _, _ = _data(n_content=64)
    bo, bd = bo.cuda(), bd.cuda()
    msgs = [torch.from_numpy(cf.messages(32)[k]) for k in (2, 1)]
    gvec = torch.rand(32, 6, 6, 3, device="cuda")

    def render(message):
        for e in m.msg_encoder.embeddings:
            e.weight.grad = None
        out = m.render(bo, bd, message, **KW)
        (out["image"] * gvec).sum().backward()
        bits = [int(v) for v in message]
        return out["image"].detach().clone(), out["depth"].detach().clone(), m.msg_encoder.embeddings[bits[0]].weight.grad.clone()

    want = [render(msg) for msg in msgs]
    with torch.no_grad():
        clean = m.render(bo, bd, None, **KW)["image"].clone()
    from nerf_signature_amd.raymarching import padded_point_count
    n_points = int(m.step_counter[(m.local_step - 1) % 16, 0])
    rec = m.fix_rays(bo, bd, dt_gamma=0, max_steps=1024)              # no point_capacity: sized by one counting march
    assert int(rec["counter"][0]) == n_points and rec["capacity"] == padded_point_count(n_points) and rec["fixed"].refreshes == 1
    assert getattr(m, "point_capacity", None) is None                  # other renders with as many rays are not capped by it
    for msg, (img, depth, grad) in zip(msgs, want):
        got = render(msg)
        assert torch.equal(got[0], img) and torch.allclose(got[1], depth, rtol=0, atol=0, equal_nan=True)
        # the gradient goes through the kept scatter plan (the slice-binned fixed-point route) instead of the per-launch record route:
        # the same sums in another order
        assert float((got[2] - grad).norm() / grad.norm()) < 1e-5
    with torch.no_grad():
        assert torch.equal(m.render(bo, bd, None, **KW)["image"], clean)           # no message: base planes only
    assert rec["fixed"].refreshes == 1                                              # nothing was recomputed for any of this

    # a base table changes (stage-1 weights loaded): noticed at the next render, planes refreshed in place
    planes_ptr = rec["fixed"].planes.data_ptr()
    with torch.no_grad():
        m.encoder.embeddings[15].weight.mul_(0.5)
        want2 = m.render(bo.clone(), bd.clone(), msgs[0], **KW)["image"].clone()    # (fresh tensors: the ordinary path)
        got2 = m.render(bo, bd, msgs[0], **KW)["image"]
    assert rec["fixed"].refreshes == 2 and rec["fixed"].planes.data_ptr() == planes_ptr
    assert torch.equal(got2, want2) and not torch.equal(got2, want[0][0])

    # the occupancy grid changes: re-marched and refreshed in place
    with torch.no_grad():
        grid = m.density_grid.clone()
        grid[0, ::2] = 0
        m.density_grid.copy_(grid)
        from nerf_signature_amd import raymarching
        m.density_bitfield.copy_(raymarching.packbits(m.density_grid, 10.0))
        want3 = m.render(bo.clone(), bd.clone(), msgs[0], **KW)["image"].clone()
        got3 = m.render(bo, bd, msgs[0], **KW)["image"]
    assert rec["fixed"].refreshes == 3 and torch.equal(got3, want3) and not torch.equal(got3, got2)
    # ... and a grid with MORE occupied cells than the buffers were sized for: sized again (new buffers), nothing dropped
    with torch.no_grad():
        m.density_grid.fill_(100.0)
        m.density_bitfield.copy_(raymarching.packbits(m.density_grid, 10.0))
        want3b = m.render(bo.clone(), bd.clone(), msgs[0], **KW)["image"].clone()
        got3b = m.render(bo, bd, msgs[0], **KW)["image"]
        rec = next(r for r in m._marched.values() if r.get("fixed") is not None)
    assert rec["capacity"] > padded_point_count(n_points) and int(rec["counter"][0]) <= rec["capacity"] and torch.equal(got3b, want3b)
    refreshes = rec["fixed"].refreshes

    # the model re-packs its own grid (update_extra_state -> packbits writes the bitfield IN PLACE through its raw pointer: neither the
    # address nor -- without mark_dirty -- the version would move): the explicit grid epoch and the dirty mark make the kept samples notice
    with torch.no_grad():
        key_before, version_before = m.grid_key(), m.density_bitfield._version
        ptr_before = m.density_bitfield.data_ptr()
        m.iter_density = 0
        m.density_grid.zero_()                     # (the update then leaves the field's own densities: a grid unlike the all-occupied one)
        m.update_extra_state(message=None)
        assert m.density_bitfield.data_ptr() == ptr_before and m.grid_key() != key_before and m.density_bitfield._version > version_before
        want3c = m.render(bo.clone(), bd.clone(), msgs[0], **KW)["image"].clone()
        got3c = m.render(bo, bd, msgs[0], **KW)["image"]
        rec = next(r for r in m._marched.values() if r.get("fixed") is not None)
    assert rec["fixed"].refreshes > refreshes and torch.equal(got3c, want3c) and not torch.equal(got3c, got3b)
    refreshes = rec["fixed"].refreshes

    # the rays change in place: they drop out of the cache by themselves (matched by address AND version)
    with torch.no_grad():
        bd.copy_(torch.nn.functional.normalize(bd + 0.01, dim=-1))
        want4 = m.render(bo.clone(), bd.clone(), msgs[0], **KW)["image"].clone()
        got4 = m.render(bo, bd, msgs[0], **KW)["image"]
    assert rec["fixed"].refreshes == refreshes and torch.equal(got4, want4)


def test_fixed_block_cache_trains_like_the_loop_that_recomputes():
    """GraphedWatermarkLoop(fixed_blocks=True) against fixed_blocks=False: same messages, block rays replaced twice on the way
    (`data` at the step itself, `next_data` one step early), a checkpoint-style invalidate in between."""
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam

    def make(seed):
        bo, bd, co, cd, gt = _data(n_content=300, seed=seed, shift=3 * seed)
        return {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()},
                "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": (gt * (0.4 + 0.15 * seed)).cuda()}}

    datas = [make(s) for s in range(3)]
    msgs = [torch.from_numpy(np.random.RandomState(20 + s).randint(0, 2, 32).astype(np.float32)) for s in range(7)]
    kw = dict(dt_gamma=0, max_steps=1024)
    runs = {}
    for fixed in (False, True):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=True)
        loop = trainer.GraphedWatermarkLoop(m, opt, kw, datas[0], headroom=0.5, fixed_blocks=fixed)
        held = []
        for k, msg in enumerate(msgs):
            nxt = msgs[k + 1] if k + 1 < len(msgs) else None
            if k == 2:
                out = loop.step(msg, data={"watermark": datas[1]["watermark"]}, next_message=nxt)            # new block rays, at the step itself
            elif k == 3:
                out = loop.step(msg, next_data={"watermark": datas[2]["watermark"]}, next_message=nxt)       # ... and one step early
            else:
                if k == 5:
                    loop.invalidate()
                out = loop.step(msg, next_message=nxt)
            held.append([v.detach().clone() for v in out[3:6]])
            if k == 0:
                first_counts = loop.point_counts()
        torch.cuda.synchronize()
        assert not loop.overflowed() and loop.point_counts()[0] != first_counts[0]      # the replaced block rays were really marched
        runs[fixed] = ([[float(v) for v in row] for row in held], torch.cat([e.weight.detach().reshape(-1) for e in m.msg_encoder.embeddings]),
                       loop.point_counts())
        if fixed:
            rec = loop.marched[0]
            assert rec["fixed"].refreshes == 4          # prepare, step 2 (data), step 4 (next_data of step 3), invalidate
            assert len(loop.marched) == 1
    (l0, t0, n0), (l1, t1, n1) = runs[False], runs[True]
    assert n0 == n1
    np.testing.assert_allclose(l1, l0, rtol=1e-4, atol=1e-6)
    start = torch.cat([torch.from_numpy(cf.table(100 + l, scale=0.05)).reshape(-1) for l in range(64)]).cuda()
    assert float((t1 - t0).norm()) <= 0.02 * float((t0 - start).norm())


def test_kept_forward_equals_the_ordinary_route():
    """FixedPoints + hg_encode_codebook_plane + field_fwd(planes) against the ordinary 17-level route: sigma, rgb and the ReLU masks bit for bit; without a message the base
    planes alone."""
    from nerf_signature_amd import fieldops as fo
    m, _, _ = _model()
    rng = np.random.RandomState(3)
    n = 20000 + 7                                        # not a multiple of 32: the padded tail of the last tile
    xyzs = torch.from_numpy((rng.rand(n, 3) * 1.2 - 0.6).astype(np.float32)).cuda()
    xyzs[:3] = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [0.0, 0.0, 0.0]])      # box corners, cell boundaries
    dirs = torch.nn.functional.normalize(torch.from_numpy(rng.randn(n, 3).astype(np.float32)), dim=-1).cuda()
    base, packed = m.encoder.tables(), m._packed()
    S = fo.codebook_presum(fo.select_tables(m.msg_encoder.tables(), tuple(int(v) for v in cf.messages(32)[2])))
    want = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=True)
    kept = fo.FixedPoints(xyzs, 1.0, base)
    got = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, fixed=kept)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[3], want[3])
    clean = fo.field_forward(xyzs, dirs, 1.0, base, None, packed, fixed=kept)          # no codebook: base planes only
    assert torch.equal(clean[0], fo.field_forward(xyzs, dirs, 1.0, base, None, packed, planes=True)[0])


def test_fixed_rays_under_autocast_and_a_scaled_loss():
    """The reference's Trainer wraps train_step in autocast(fp16) and scales the loss by 65536 (utils_wtmk_disen.py:1172-1178): the kept route
    under both equals the recomputing route under both (the kept scatter plan's largest-gradient word is cleared every step, so a step with
    a 65536-fold gradient does not coarsen the fixed-point scale of the next one)."""
    m, _, _ = _model()
    bo, bd, _, _, _ = _data(n_content=64)
    bo, bd = bo.cuda(), bd.cuda()
    msg = torch.from_numpy(cf.messages(32)[2])
    gvec = torch.rand(32, 6, 6, 3, device="cuda")
    tab = m.msg_encoder.embeddings[int(msg[0])].weight

    def run(o, d, scale):
        tab.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            out = m.render(o, d, msg, **KW)
        ((out["image"].float() * gvec).sum() * scale).backward()
        return out["image"].detach().clone(), tab.grad.clone()

    want_big, want_one = run(bo.clone(), bd.clone(), 65536.0), run(bo.clone(), bd.clone(), 1.0)
    rec = m.fix_rays(bo, bd, dt_gamma=0, max_steps=1024)
    got_big, got_one = run(bo, bd, 65536.0), run(bo, bd, 1.0)          # the large step first: its scale must not leak into the next
    assert rec["fixed"].refreshes == 1
    for got, want in ((got_big, want_big), (got_one, want_one)):
        assert got[0].dtype == want[0].dtype and torch.equal(got[0], want[0])
        assert bool(torch.isfinite(got[1]).all()) and float((got[1] - want[1]).norm() / want[1].norm()) < 1e-5
    assert float((got_big[1] / 65536.0 - got_one[1]).norm() / got_one[1].norm()) < 1e-4


def test_fixed_blocks_loop_notices_a_base_table_written_between_replays():
    """A captured replay runs no Python, so the kept planes cannot notice a changed base table by themselves: step() compares the versions
    of the 16 base tables and the occupancy bitfield with those the blocks were fixed under and re-fixes before the replay."""
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam
    bo, bd, co, cd, gt = _data(n_content=300)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    msgs = [torch.from_numpy(np.random.RandomState(60 + s).randint(0, 2, 32).astype(np.float32)) for s in range(4)]
    runs = {}
    for fixed in (False, True):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=True)
        loop = trainer.GraphedWatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024), data, fixed_blocks=fixed)
        held = []
        for k, msg in enumerate(msgs):
            if k == 2:
                with torch.no_grad():
                    m.encoder.embeddings[14].weight.mul_(0.25)           # e.g. new stage-1 weights copied in, no invalidate() call
            held.append(loop.step(msg)[3:6])
            held[-1] = [v.detach().clone() for v in held[-1]]
        torch.cuda.synchronize()
        runs[fixed] = [[float(v) for v in row] for row in held]
        if fixed:
            assert loop.marched[0]["fixed"].refreshes == 2
    np.testing.assert_allclose(runs[True], runs[False], rtol=1e-4, atol=1e-6)
    assert abs(runs[False][2][1] - runs[False][1][1]) > 1e-6
