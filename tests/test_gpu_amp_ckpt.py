"""The reference Trainer's AMP loop around this repo's train_step, and the checkpoint shim on the GPU.

AMP (utils_wtmk_disen.py:1172-1178, `-O` => fp16=True, main_nerf_wtmk.py:79-82): `with autocast(fp16): train_step(...)`,
`scaler.scale(loss).backward()`, `scaler.step(optimizer)`, `scaler.update()` with a plain torch.optim.Adam over get_params() -- the
reference's own objects, no GradSink.  Checked: the update equals the fp32 loop's within tolerance (every native entry point keeps
fp32 arithmetic under autocast; 65536x gradients pass through k_field_bwd's fixed-point scatter and are unscaled on the fanned-out
table gradients), and a non-finite loss skips the step and halves the scale.

N4: save -> load -> bit-identical render; a checkpoint loaded between two replays of the captured loop."""
import copy
import os

import numpy as np
import pytest
import torch

import closed_form as cf
from test_gpu_render import _data, _model

pytestmark = pytest.mark.gpu
KW = dict(dt_gamma=0, max_steps=1024)


def _cuda_data(n_content=400, block=6, big=False):
    bo, bd, co, cd, gt = _data(n_content=n_content, block=block)
    return {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}


def _snapshot(m):
    return [e.weight.detach().clone() for e in m.msg_encoder.embeddings], torch.cat([p.detach().reshape(-1) for p in m.msg_decoder.parameters()])


@pytest.mark.parametrize("n_content,block", [(400, 6), (3000, 12)])     # small: record scatter with float atomics; large: binned fixed-point scatter
def test_train_step_under_autocast_and_gradscaler(n_content, block):
    from nerf_signature_amd import trainer
    data = _cuda_data(n_content, block)
    msgs = [torch.from_numpy(np.random.RandomState(60 + s).randint(0, 2, 32).astype(np.float32)).cuda() for s in range(3)]
    results = []
    for amp in (False, True):
        torch.manual_seed(0)
        m, _, _ = _model()
        m.grad_sink = None
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", enabled=amp)           # init_scale 65536
        losses = []
        for msg in msgs:
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                out = trainer.train_step(m, data, msg, dict(KW, fp16=True, workspace="w"))
            assert out[2].dtype == torch.float32 and out[5].dtype == torch.float32        # fp32 arithmetic behind the C ABI under autocast
            scaler.scale(out[5]).backward()
            if amp:      # the scaled gradient really went through the native backward
                sel = [e.weight for e in m.msg_encoder.embeddings if e.weight.grad is not None]
                assert len(sel) == 32 and all(torch.isfinite(t.grad).all() for t in sel)
            scaler.step(opt)
            scaler.update()
            losses.append([float(v.detach()) for v in out[3:6]])
        results.append((np.array(losses), *_snapshot(m), scaler.get_scale() if amp else 1.0))
    (l0, t0, d0, _), (l1, t1, d1, scale) = results
    assert scale == 65536.0                                           # no overflow: the scale was never reduced
    np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=2e-5)
    init = [torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda() for l in range(64)]
    moved = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, init)) ** 0.5
    diff = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, t1)) ** 0.5
    assert moved > 0 and diff / moved < 0.05                          # (Adam turns last-bit differences of tiny gradients into +-lr steps)
    assert float((d0 - d1).norm() / d0.norm()) < 0.05
    # unselected tables never moved on either side
    bits = [[int(v) for v in msg.cpu()] for msg in msgs]
    for l in range(64):
        if all(b[l // 2] != l % 2 for b in bits):
            assert torch.equal(t1[l], init[l]) and torch.equal(t0[l], init[l])


@pytest.mark.parametrize("n_content,block", [(400, 6), (3000, 12)])
def test_gradscaler_skips_a_non_finite_step(n_content, block):
    """Overflow handling of `scaler.step` (utils_wtmk_disen.py:1175-1178): a scaled loss that overflows must leave non-finite values in
    the gradients the scaler inspects -- including the table gradients that come out of the integer scatter -- so the step is skipped,
    no parameter moves, and the scale is halved; the next (finite) step trains again."""
    from nerf_signature_amd import trainer
    data = _cuda_data(n_content, block)
    m, _, _ = _model()
    m.grad_sink = None
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    msg = torch.from_numpy(cf.messages(32)[2]).cuda()
    before = _snapshot(m)
    bad = copy.deepcopy(data)
    bad["content"]["images"].fill_(3e37)                              # d(MSE)/d(pred) overflows fp32 once scaled by 65536: inf in the content render's gradient only
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        out = trainer.train_step(m, bad, msg, KW)
    scaler.scale(out[5]).backward()
    grads = [e.weight.grad for e in m.msg_encoder.embeddings if e.weight.grad is not None]
    assert len(grads) == 32 and not bool(torch.isfinite(grads[0]).all())   # the table gradient itself shows the overflow
    scaler.step(opt)
    scaler.update()
    after = _snapshot(m)
    assert all(torch.equal(a, b) for a, b in zip(before[0], after[0])) and torch.equal(before[1], after[1])
    assert scaler.get_scale() == 32768.0
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        out = trainer.train_step(m, data, msg, KW)
    scaler.scale(out[5]).backward()
    scaler.step(opt)
    scaler.update()
    assert not torch.equal(before[1], _snapshot(m)[1]) and scaler.get_scale() == 32768.0 and bool(torch.isfinite(out[5]))


def test_checkpoint_save_load_renders_bit_identically(tmp_path):
    """N4 on the GPU: a trained model saved in the reference's dict format and loaded into a fresh model renders the same bits
    (training-mode and eval-mode paths, with and without a message); a clean stage-1 checkpoint (no codebook/decoder keys) loads with
    strict=False and renders the clean image."""
    from nerf_signature_amd import checkpoint as ck, trainer
    data = _cuda_data()
    m, _, _ = _model()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = trainer.WatermarkLoop(m, opt, KW)
    msg = torch.from_numpy(cf.messages(32)[2])
    for _ in range(2):
        loop.step(data, msg)
    path = ck.save_checkpoint(str(tmp_path / "checkpoints" / "ngp_ep0002.pth"), m, epoch=2, global_step=2, optimizer=opt, full=True)
    fresh, _, _ = _model()
    with torch.no_grad():
        for p in fresh.msg_decoder.parameters():
            p.add_(0.01)
    opt2 = torch.optim.Adam(fresh.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    missing, unexpected, meta = ck.load_checkpoint(path, fresh, optimizer=opt2, map_location="cuda")
    assert missing == [] and unexpected == [] and meta["epoch"] == 2 and "skipped" not in meta
    o, d = data["content"]["rays_o"], data["content"]["rays_d"]
    for model_msg in (msg, None):
        for mode in ("train", "eval"):
            getattr(m, mode)(), getattr(fresh, mode)()
            with torch.no_grad():
                a = m.render(o, d, model_msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **KW)
                b = fresh.render(o, d, model_msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **KW)
            assert torch.equal(a["image"], b["image"]) and torch.equal(torch.nan_to_num(a["depth"]), torch.nan_to_num(b["depth"]))
    m.train(), fresh.train()
    # the restored optimiser continues identically
    la = loop.step(data, msg)[5].detach().clone()
    lb = trainer.WatermarkLoop(fresh, opt2, KW).step(data, msg)[5].detach().clone()
    np.testing.assert_allclose(float(la), float(lb), rtol=1e-5)
    clean = {k: v for k, v in m.state_dict().items() if not k.startswith("msg_")}
    clean["sigma_net.params"] = clean["sigma_net.params"].half()
    target, _, _ = _model()
    missing, unexpected, _ = ck.load_checkpoint({"model": clean, "epoch": 1, "global_step": 1, "stats": {}}, target, model_only=True)
    assert unexpected == [] and all(k.startswith("msg_") for k in missing)


def test_checkpoint_loaded_between_replays_of_the_captured_loop():
    """load_checkpoint on a model that a GraphedWatermarkLoop has captured.  The pre-sum buffer and the optimiser state
    tensors are held by the graph by address: they must survive the load in place, the announced-message bookkeeping must be reset,
    and the following replays must equal an eager loop that loaded the same checkpoint."""
    from nerf_signature_amd import checkpoint as ck, trainer
    from nerf_signature_amd.optim import CodebookAdam
    data = _cuda_data(n_content=300)
    msgs = [torch.from_numpy(np.random.RandomState(s).randint(0, 2, 32).astype(np.float32)) for s in range(5)]
    # the checkpoint: two eager steps from the initial state
    torch.manual_seed(0)
    src, _, _ = _model()
    opt_src = CodebookAdam(src.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop_src = trainer.WatermarkLoop(src, opt_src, KW)
    for k in range(2):
        loop_src.step(data, msgs[k])
    state = copy.deepcopy(ck.checkpoint_state(src, epoch=1, global_step=2, optimizer=opt_src, full=True))
    runs = []
    for graphed in (False, True):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=graphed)
        if graphed:
            loop = trainer.GraphedWatermarkLoop(m, opt, KW, data)
            step = lambda k: loop.step(msgs[k], next_message=msgs[k + 1] if k + 1 < len(msgs) else None)
        else:
            loop = trainer.WatermarkLoop(m, opt, KW)
            step = lambda k: loop.step(data, msgs[k])
        step(0)                                   # diverge from the checkpoint's history ...
        if graphed:
            s_buf = m._presum_cache[1].data_ptr()
            state_ptrs = {id(p): opt.state[p]["exp_avg"].data_ptr() for p in m.msg_encoder.parameters() if len(opt.state[p])}
        missing, unexpected, meta = ck.load_checkpoint(copy.deepcopy(state), m, optimizer=opt)    # ... then resume from it
        assert missing == [] and unexpected == [] and "skipped" not in meta
        if graphed:
            assert m._presum_cache[1].data_ptr() == s_buf and m._presum_cache[0] is None and loop._s_for is None
            assert all(opt.state[p]["exp_avg"].data_ptr() == state_ptrs[id(p)] for p in m.msg_encoder.parameters() if id(p) in state_ptrs)
            assert torch.is_tensor(opt.param_groups[0]["lr"]) and opt.param_groups[0]["lr"].is_cuda
        losses = [float(step(k)[5].detach()) for k in (2, 3, 4)]
        torch.cuda.synchronize()
        if graphed:
            assert not loop.overflowed()
        runs.append((losses, _snapshot(m)))
    (l0, (t0, d0)), (l1, (t1, d1)) = runs
    np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=2e-4)
    init = [torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda() for l in range(64)]
    moved = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, init)) ** 0.5
    diff = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, t1)) ** 0.5
    assert moved > 0 and diff / moved < 0.05


def test_eager_loop_refuses_a_model_left_in_device_select_mode():
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam
    data = _cuda_data(n_content=300)
    m, _, _ = _model()
    opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=True)
    g = trainer.GraphedWatermarkLoop(m, opt, KW, data)
    msg = torch.from_numpy(cf.messages(32)[2])
    g.step(msg)
    eager = trainer.WatermarkLoop(m, opt, KW)
    with pytest.raises(RuntimeError, match="device-select"):
        eager.step(data, msg.cuda())
    g.close()
    eager.step(data, msg)          # host-side selection again: only the 32 selected tables are recorded in the sink
    assert len(eager.sink.selected) == 32


def test_capacity_overflow_is_detected_and_recaptured():
    """an overflowing replay (more sample points than the captured buffers hold) drops rays like the reference's bounded
    mode and used to be reported only after the fact.  ensure_capacity() re-sizes from the current rays and captures again; the next
    step is whole again and equals the eager loop's."""
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam
    data = _cuda_data(n_content=300)
    dense = {"watermark": data["watermark"],      # content rays that all cross the ball: several times the points the capture was sized for
             "content": {"rays_o": data["watermark"]["rays_o_block"].reshape(1, -1, 3)[:, :300].contiguous(),
                         "rays_d": data["watermark"]["rays_d_block"].reshape(1, -1, 3)[:, :300].contiguous(), "images": data["content"]["images"]}}
    msg = torch.from_numpy(cf.messages(32)[2])
    torch.manual_seed(0)
    m, _, _ = _model()
    opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=True)
    loop = trainer.GraphedWatermarkLoop(m, opt, KW, data)
    loop.step(msg)
    assert not loop.overflowed() and not loop.ensure_capacity()
    loop.step(msg, data={"content": dense["content"]})
    assert loop.overflowed()
    cap_before = loop.content_capacity
    assert loop.ensure_capacity() and loop.content_capacity > cap_before
    out = loop.step(msg)
    torch.cuda.synchronize()
    assert not loop.overflowed()
    torch.manual_seed(0)
    ref_m, _, _ = _model()
    ref_m.load_state_dict(m.state_dict())
    with torch.no_grad():
        want = trainer.train_step(ref_m, dense, msg, KW)
    # (the parameters moved by one more optimiser step between the two evaluations: compare the image loss loosely, the point is that
    #  no ray is dropped any more: the loss of a render with dropped rays is off by orders of magnitude)
    assert abs(float(out[3]) - float(want[3])) < 0.05 * float(want[3]) + 1e-4


def test_shared_gradient_step_under_a_foreign_loop_matches_plain_autograd_and_adam():
    """What the drop-in modules switch on for the reference's own Trainer (NeRFNetwork.shared_gradient_step + auto_fix_rays): zero_grad / train_step under
    autocast / GradScaler.scale(loss).backward() / scaler.step(torch.optim.Adam) / scaler.update(), four steps with a new message each -- against the same
    loop on a model with plain autograd gradients (D dense fan-outs, the optimiser's own loop).  Same losses, same parameters (the fused pass IS
    torch.optim.Adam's arithmetic), same per-table Adam step counts in torch's own state format, a skipped step (injected inf) skipped on both sides, and
    between backward() and step() exactly ONE selected table carries a `.grad` -- the shared tensor -- which GradScaler unscales once."""
    import test_gpu_render as T
    from nerf_signature_amd import trainer
    bo, bd, co, cd, gt = T._data(n_content=300)
    msgs = [torch.from_numpy(np.random.RandomState(s).randint(0, 2, 32).astype(np.float32)).cuda() for s in range(4)]
    kw = dict(dt_gamma=0, max_steps=1024)
    runs = []
    for shared in (False, True):
        torch.manual_seed(0)
        m, _, _ = T._model()
        m.shared_gradient_step = shared
        m.auto_fix_rays = shared
        data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        losses, carriers = [], []
        for k, msg in enumerate(msgs):
            opt.zero_grad()
            with torch.autocast("cuda"):
                out = trainer.train_step(m, data, msg, kw, lambda_w=0.005, lambda_i=1.0)
            loss = out[5] * (float("inf") if k == 2 else 1.0)           # step 2 overflows: GradScaler must skip it and halve the scale, on both sides
            scaler.scale(loss).backward()
            with_grad = [i for i, e in enumerate(m.msg_encoder.embeddings) if e.weight.grad is not None]
            carriers.append(len(with_grad))
            scaler.step(opt)
            scaler.update()
            losses.append(float(out[5].detach()))
        torch.cuda.synchronize()
        tables = [e.weight.detach().clone() for e in m.msg_encoder.embeddings]
        steps = [float(opt.state[e.weight]["step"]) if len(opt.state[e.weight]) else 0.0 for e in m.msg_encoder.embeddings]
        dec = torch.cat([p.detach().reshape(-1) for p in m.msg_decoder.parameters()])
        # the decoder's dense gradients: stepped by the optimiser itself (plain) or by the hook's one-launch pass (shared) -- three real steps in torch's own state
        # format either way, and the gradients still readable after optimizer.step()
        stepped = [p for p in m.msg_decoder.parameters() if len(opt.state[p])]
        assert len(stepped) >= 27 and {float(opt.state[p]["step"]) for p in stepped} == {3.0} and not any(opt.state[p]["step"].is_cuda for p in stepped)
        assert all(p.grad is not None for p in stepped)
        runs.append((losses, tables, steps, dec, carriers, scaler.get_scale(), m))
    (l0, t0, s0, d0, c0, sc0, m0), (l1, t1, s1, d1, c1, sc1, m1) = runs
    assert c0 == [32, 32, 32, 32] and c1 == [1, 1, 1, 1]                    # plain: every selected table has a dense gradient; shared: the carrier alone
    assert sc0 == sc1 == 512.0                                              # one skipped step on both sides
    assert s0 == s1 and sum(s0) == 3 * 32                                   # three real steps, per-table counts as torch.optim.Adam keeps them
    np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=2e-5)
    moved = sum(float((a - torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda()).pow(2).sum()) for l, a in enumerate(t0)) ** 0.5
    diff = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, t1)) ** 0.5
    assert moved > 0 and diff / moved < 0.05                                # (Adam with eps = 1e-15 turns last-bit differences of G into +-lr steps: aggregate bound)
    assert float((d0 - d1).norm() / d0.norm()) < 0.05
    assert all(e.weight.grad is None for e in m1.msg_encoder.embeddings)    # consumed by the fused pass
    # the block rays were seen twice: the kept-planes route is in use from the second step on (same renders bit for bit: tests/test_gpu_fixed.py)
    assert any(r.get("fixed") is not None for r in getattr(m1, "_marched", {}).values()) and not getattr(m0, "_marched", None)
    # an optimiser the fused pass does not implement (weight decay): the shared gradient dissolves into ordinary dense gradients, the step is torch's
    opt_w = torch.optim.Adam(m1.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, weight_decay=1e-4)
    opt_w.zero_grad()
    out = trainer.train_step(m1, data, msgs[0], kw)
    out[5].backward()
    assert sum(e.weight.grad is not None for e in m1.msg_encoder.embeddings) == 1
    before = [e.weight.detach().clone() for e in m1.msg_encoder.embeddings]
    opt_w.step()
    assert sum(int(not torch.equal(a, e.weight.detach())) for a, e in zip(before, m1.msg_encoder.embeddings)) == 32
    assert sum(e.weight.grad is not None for e in m1.msg_encoder.embeddings) == 32


@pytest.mark.parametrize("distortion,loss_w", [("none", "bce"), ("brightness", "bce"), ("none", "mse")])
def test_reference_trainer_train_step_equals_the_stock_operator_sequence(distortion, loss_w):
    """trainer.reference_trainer_train_step -- what the drop-in directory binds as Trainer.train_step -- against the operator sequence the reference's
    own method issues around model.render / model.msg_decoder (utils_wtmk_disen.py:579-646: clamp, distortion layer, layout change, normalisation,
    decoder, element-wise MSE and its mean, loss_w at temperature 10, weighted sum), on a stand-in for the reference's Trainer object (the attributes
    its method reads): same six return values, same gradients."""
    import argparse
    import types
    from nerf_signature_amd import trainer
    from nerf_signature_amd.distortion import DistortionLayer
    bo, bd, co, cd, gt = _data(n_content=300)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda(), "images": torch.zeros(32, 6, 6, 3)},
            "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    msg = torch.from_numpy(np.random.RandomState(7).randint(0, 2, 32).astype(np.float32)).cuda()
    opt = argparse.Namespace(dt_gamma=0, max_steps=1024, color_space="srgb", loss_w=loss_w, distortion=distortion, workspace="w", fp16=False)
    results = []
    for fused in (False, True):
        torch.manual_seed(0)
        m, _, _ = _model()
        me = types.SimpleNamespace(model=m, opt=opt, lambda_w=0.005, lambda_i=1.0, distortion=distortion)
        if fused:
            out = trainer.reference_trainer_train_step(me, data, msg)
            if distortion != "none":
                draw = me._nsig_distortion_layer.param.clone()
        else:
            wm, ct = data["watermark"], data["content"]
            img = m.render(wm["rays_o_block"], wm["rays_d_block"], msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **vars(opt))["image"]
            pred = torch.clamp(img, min=0, max=1)
            seen = pred
            if distortion != "none":
                layer = DistortionLayer(distortion)            # same seed as the method's own layer: same first draw
                layer.draw(tuple(pred.shape), pred.device)
                seen = layer(pred)
                draw = layer.param.clone()
            decoded = m.msg_decoder(m.normalization(seen.permute(0, 3, 1, 2)))
            cpred = m.render(ct["rays_o"], ct["rays_d"], msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **vars(opt))["image"]
            lossi = torch.nn.MSELoss(reduction="none")(cpred, ct["images"]).mean()
            lossw = (trainer.loss_w_bce if loss_w == "bce" else trainer.loss_w_mse)(decoded, msg.unsqueeze(-1))
            out = (pred, ct["images"], cpred, lossi, lossw, 0.005 * lossw + 1.0 * lossi)
        out[5].backward()
        bits = [int(v) for v in msg.cpu()]
        G = m.msg_encoder.embeddings[2 * 0 + bits[0]].weight.grad.clone()
        dgrad = torch.cat([p.grad.reshape(-1) for p in m.msg_decoder.parameters() if p.grad is not None])
        results.append(([o.detach().clone() for o in out], G, dgrad, draw if distortion != "none" else None))
    (o0, G0, d0, w0), (o1, G1, d1, w1) = results
    if distortion != "none":
        assert torch.equal(w0, w1)
    for a, b in zip(o0[:3], o1[:3]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=1e-6)
    for a, b in zip(o0[3:], o1[3:]):
        np.testing.assert_allclose(float(b), float(a), rtol=2e-5, atol=1e-8)
    assert float(G0.abs().max()) > 0 and float((G1 - G0).norm() / G0.norm()) < 2e-3
    assert float((d1 - d0).norm() / d0.norm()) < 2e-3
    with pytest.raises(UnboundLocalError):      # 4-channel block images without a background model: the reference's own failure (:585-590)
        bad = dict(data, watermark=dict(data["watermark"], images=torch.zeros(32, 6, 6, 4)))
        trainer.reference_trainer_train_step(types.SimpleNamespace(model=m, opt=opt, lambda_w=1.0, lambda_i=1.0, distortion="none"), bad, msg)


@pytest.mark.parametrize("whole_step", [False, True])
@pytest.mark.parametrize("distortion", ["brightness", "noise", "rotation", "scaling"])
def test_block_graph_with_a_distortion_layer(monkeypatch, distortion, whole_step):
    """The captured block render + decoder with the step's distortion layer inside (its draws land in static device buffers before every replay): the same
    loop with and without the graphs, same draws (the layer's generators are seeded alike) -> same losses step by step.  `scaling` changes the decoder's
    input width every step: the graph steps aside (no capture) and the eager launches run."""
    import argparse
    import types
    from nerf_signature_amd import trainer
    n_steps = 12 if whole_step else 8           # (the whole-step capture sizes its sample buffers first: four more eager steps)
    bo, bd, _, _, _ = _data(n_content=300)
    wm = {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda(), "images": torch.zeros(32, 6, 6, 3)}
    batches = []
    for k in range(n_steps):
        _, _, co, cd, gt = _data(n_content=300, seed=30 + k)
        batches.append({"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()})
    gen = torch.Generator(device="cuda").manual_seed(6)
    msgs = [torch.randint(0, 2, (32,), generator=gen, device="cuda").float() for _ in range(n_steps)]
    opt_ns = argparse.Namespace(dt_gamma=0, max_steps=1024, color_space="srgb", loss_w="bce", distortion=distortion, workspace="w", fp16=False)
    runs = []
    for graph_on in (False, True):
        monkeypatch.setenv("NERFSIG_DROPIN_OFF", ",".join(([] if whole_step else ["step_graph"]) + ([] if graph_on else ["block_graph"])))
        torch.manual_seed(0)
        m, _, _ = _model()
        m.shared_gradient_step = m.auto_fix_rays = True
        me = types.SimpleNamespace(model=m, opt=opt_ns, lambda_w=0.005, lambda_i=1.0, distortion=distortion)
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        losses = []
        for k in range(n_steps):
            opt.zero_grad()
            out = trainer.reference_trainer_train_step(me, {"watermark": wm, "content": batches[k]}, msgs[k])
            out[5].backward()
            opt.step()
            losses.append([float(v.detach()) for v in out[3:6]])
        runs.append((np.array(losses), me))
    (l0, _), (l1, me1) = runs
    g = me1._nsig_block_graph
    assert g.failed is None and (g.captures, g.generation) == ((0, 0) if distortion == "scaling" else (1, 3))
    np.testing.assert_allclose(l1, l0, rtol=5e-5, atol=1e-7)
    assert len({round(float(v), 7) for v in l1[:, 1]}) == n_steps      # a new draw (and a new message) every step: the watermark loss never repeats


@pytest.mark.parametrize("whole_step", [False, True])
def test_block_graph_replays_the_eager_block_render_and_decoder(monkeypatch, whole_step):
    """blockgraph.BlockDecodeGraph under the reference Trainer's loop shape (zero_grad / autocast train_step / GradScaler / torch.optim.Adam, a new device-side
    message and new content rays every step): the bound Trainer.train_step with the block render + decoder replayed from two captured graphs against the same
    loop with NERFSIG_DROPIN_OFF=block_graph -- same losses step by step, same parameters after 10 steps, one capture, replays from the fifth step on (two sightings
    to keep the rays, three eager steps on the kept route), a step GradScaler skips skipped on both sides; then the guards: backward of a stale forward, a
    second backward, and new block rays (the graph steps aside, the eager route runs, a new capture follows)."""
    import argparse
    import types
    from nerf_signature_amd import trainer
    # whole_step: blockgraph.StepGraph captures the content render and the losses as well
    n_steps = 14 if whole_step else 10          # (the whole-step capture sizes its sample buffers first: four more eager steps)
    bo, bd, _, _, _ = _data(n_content=300)
    block_o, block_d = bo.cuda(), bd.cuda()
    batches = []
    for k in range(n_steps):
        _, _, co, cd, gt = _data(n_content=300, seed=10 + k)
        batches.append({"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()})
    gen = torch.Generator(device="cuda").manual_seed(5)
    msgs = [torch.randint(0, 2, (32,), generator=gen, device="cuda").float() for _ in range(n_steps)]
    opt_ns = argparse.Namespace(dt_gamma=0, max_steps=1024, color_space="srgb", loss_w="bce", distortion="none", workspace="w", fp16=True)
    runs = []
    for graph_on in (False, True):
        monkeypatch.setenv("NERFSIG_DROPIN_OFF", ",".join(([] if whole_step else ["step_graph"]) + ([] if graph_on else ["block_graph"])))
        torch.manual_seed(0)
        m, _, _ = _model()
        m.shared_gradient_step = m.auto_fix_rays = True
        me = types.SimpleNamespace(model=m, opt=opt_ns, lambda_w=0.005, lambda_i=1.0, distortion="none")
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        wm = {"rays_o_block": block_o, "rays_d_block": block_d, "images": torch.zeros(32, 6, 6, 3)}
        losses = []
        for k in range(n_steps):
            opt.zero_grad()
            with torch.autocast("cuda"):
                out = trainer.reference_trainer_train_step(me, {"watermark": wm, "content": batches[k]}, msgs[k])
            scaler.scale(out[5] * (float("inf") if k == n_steps - 3 else 1.0)).backward()
            scaler.step(opt)
            scaler.update()
            losses.append([float(v.detach()) for v in out[3:6]] + [float(out[0].sum())])
        torch.cuda.synchronize()
        runs.append((np.array(losses), [e.weight.detach().clone() for e in m.msg_encoder.embeddings],
                     torch.cat([p.detach().reshape(-1) for p in m.msg_decoder.parameters()]), scaler.get_scale(), me, m, opt))
    (l0, t0, d0, s0, me0, _, _), (l1, t1, d1, s1, me1, m1, opt1) = runs
    g = me1._nsig_block_graph
    assert me0._nsig_block_graph.captures == 0 and g.captures == 1 and g.failed is None and g.generation == 5 and s0 == s1 == 512.0
    # (run-to-run: the order of the float atomics in G; the eager route also seeds the decoder's backward from its head kernel, the graph from the loss kernel)
    np.testing.assert_allclose(l1, l0, rtol=2e-5, atol=1e-7)
    moved = sum(float((a - torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda()).pow(2).sum()) for l, a in enumerate(t0)) ** 0.5
    diff = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, t1)) ** 0.5
    assert moved > 0 and diff / moved < 0.05 and float((d0 - d1).norm() / d0.norm()) < 0.05
    # guards
    monkeypatch.setenv("NERFSIG_DROPIN_OFF", "" if whole_step else "step_graph")
    data = {"watermark": {"rays_o_block": block_o, "rays_d_block": block_d, "images": torch.zeros(32, 6, 6, 3)}, "content": batches[0]}
    opt1.zero_grad()
    first = trainer.reference_trainer_train_step(me1, data, msgs[0])
    second = trainer.reference_trainer_train_step(me1, data, msgs[1])
    with pytest.raises(RuntimeError, match="later forward"):
        first[5].backward()
    second[5].backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        second[5].backward()
    opt1.zero_grad()
    shifted = {"watermark": {"rays_o_block": block_o.clone(), "rays_d_block": block_d.clone(), "images": torch.zeros(32, 6, 6, 3)}, "content": batches[0]}
    before = g.generation
    for k in range(6):      # new ray tensors: two sightings + three eager steps, then a second capture
        opt1.zero_grad()
        out = trainer.reference_trainer_train_step(me1, shifted, msgs[k])
        out[5].backward()
        opt1.step()
    assert g.captures == 2 and g.generation == before + 1 and np.isfinite(float(out[5].detach()))
    if whole_step:
        # more samples than the captured buffers hold: THAT step runs eagerly (same losses as the eager loop: nothing is dropped), the next capture is larger
        with pytest.raises(NotImplementedError, match="lossi"):
            trainer.reference_trainer_train_step(me1, shifted, msgs[0])[3].backward()
        opt1.zero_grad()
        g.capacity, g.key, g.forward_graph = 128, None, None
        seen = []
        for k in range(9):
            opt1.zero_grad()
            out = trainer.reference_trainer_train_step(me1, shifted, msgs[k])
            out[5].backward()
            seen.append((g.overflows, g.capacity, g.captures, float(out[3].detach())))
            opt1.step()
        assert seen[-1][0] == 1 and seen[-1][1] > 128 and seen[-1][2] == 3 and all(np.isfinite(v[3]) and v[3] > 0 for v in seen)


def test_dense_takeover_is_torch_adam_arithmetic():
    """optim._dense_takeover (opt_adam_dense_host: the hook's one-launch Adam over every remaining dense gradient) against torch.optim.Adam on the same
    parameters and gradients, 5 steps, 40 tensors of ragged sizes (two launches: 32 + 8), two groups with different learning rates, a parameter without a
    gradient in one step; the gradients are set aside while the optimiser's own loop runs and are back afterwards; a group with weight decay is left alone."""
    from nerf_signature_amd import optim
    g = torch.Generator(device="cuda").manual_seed(3)
    sizes = [(1,), (7,), (64, 3, 3, 3), (1025,), (64,), (2048, 3)] * 6 + [(5, 5), (333,), (4096,), (1024,)]

    def make():
        gg = torch.Generator(device="cuda").manual_seed(3)
        return [torch.nn.Parameter(torch.randn(*s, device="cuda", generator=gg)) for s in sizes]

    a, b = make(), make()
    kw = dict(betas=(0.9, 0.99), eps=1e-15)
    oa = torch.optim.Adam([{"params": a[:25], "lr": 1e-2}, {"params": a[25:], "lr": 3e-4}], **kw)
    ob = torch.optim.Adam([{"params": b[:25], "lr": 1e-2}, {"params": b[25:], "lr": 3e-4}], **kw)
    for k in range(5):
        grads = [torch.randn(*s, device="cuda", generator=g) * 10.0 ** (k - 2) for s in sizes]
        for p, q, gr in zip(a, b, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if k == 3:
            a[7].grad = b[7].grad = None            # a parameter without a gradient this step keeps its count, like torch
        oa.step()
        with torch.no_grad():
            optim._dense_takeover(ob)
        assert all(q.grad is None for q in b)
        ob.step()                                   # nothing left for the optimiser's own loop: it must not move anything nor count a step
        optim._dense_takeover_post_step(ob, (), {})
        assert all((q.grad is None) == (k == 3 and i == 7) for i, q in enumerate(b))
    torch.cuda.synchronize()
    for i, (p, q) in enumerate(zip(a, b)):
        assert float(oa.state[p]["step"]) == float(ob.state[q]["step"]) == (4.0 if i == 7 else 5.0)
        torch.testing.assert_close(q, p, rtol=2e-6, atol=2e-7)
        # (torch updates the first moment as lerp(m, g, 1 - beta1), the kernel as beta1 * m + (1 - beta1) * g: last-bit differences where the two nearly cancel)
        scale = float(oa.state[p]["exp_avg"].abs().max())
        torch.testing.assert_close(ob.state[q]["exp_avg"], oa.state[p]["exp_avg"], rtol=2e-5, atol=1e-6 * scale)
        torch.testing.assert_close(ob.state[q]["exp_avg_sq"], oa.state[p]["exp_avg_sq"], rtol=2e-5, atol=1e-12)
    w = make()
    ow = torch.optim.Adam(w, lr=1e-2, weight_decay=1e-3)
    for p in w:
        p.grad = torch.ones_like(p)
    with torch.no_grad():
        optim._dense_takeover(ow)
    assert all(p.grad is not None for p in w) and len(ow.state) == 0


def test_fused_adam_passes_follow_a_state_dict_loaded_between_steps():
    """torch's optimizer.load_state_dict() (what the reference Trainer calls, utils_wtmk_disen.py:1500) replaces the inner state dicts AND
    their tensors; the cached handles of the two hook passes (optim.fused_shared_step for the tables, optim._dense_takeover for dense gradients) must
    notice and follow the loaded moments instead of updating the orphaned ones.  step -> load_state_dict(deepcopy(state_dict)) -> step, against plain
    torch.optim.Adam doing the same; afterwards state_dict() carries the moments that were really used (they move with the step)."""
    import copy as _copy
    from nerf_signature_amd import optim
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    gen = torch.Generator(device="cuda").manual_seed(11)
    # dense take-over
    a = [torch.nn.Parameter(torch.randn(300, 7, device="cuda", generator=gen)) for _ in range(3)]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = torch.optim.Adam(a, **kw), torch.optim.Adam(b, **kw)
    # shared-gradient table pass: 4 "tables" that all carry G
    ta = [torch.nn.Parameter(torch.randn(1 << 19, 2, device="cuda", generator=gen) * 1e-2) for _ in range(4)]
    tb = [torch.nn.Parameter(p.detach().clone()) for p in ta]
    ota, otb = torch.optim.Adam(ta, **kw), torch.optim.Adam(tb, **kw)
    for k in range(4):
        grads = [torch.randn(300, 7, device="cuda", generator=gen) for _ in a]
        G = torch.randn(1 << 19, 2, device="cuda", generator=gen)
        for p, q, g in zip(a, b, grads):
            p.grad, q.grad = g.clone(), g.clone()
        for p in ta:
            p.grad = G.clone()
        oa.step()
        ota.step()
        with torch.no_grad():
            optim._dense_takeover(ob)
            ob.step()
            optim._dense_takeover_post_step(ob, (), {})
            optim.fused_shared_step(otb, otb.param_groups[0], tb, G)
        if k == 1:      # a checkpoint comes back: new state dicts, new tensors (moments scaled so that following the wrong ones is visible)
            for o in (oa, ob, ota, otb):
                sd = _copy.deepcopy(o.state_dict())
                for st in sd["state"].values():
                    st["exp_avg"].mul_(0.5)
                    st["exp_avg_sq"].mul_(2.0)
                o.load_state_dict(sd)
    torch.cuda.synchronize()
    for ref_o, o, ref_p, ps in ((oa, ob, a, b), (ota, otb, ta, tb)):
        for p, q in zip(ref_p, ps):
            assert float(o.state[q]["step"]) == float(ref_o.state[p]["step"]) == 4.0
            torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6)
            scale = float(ref_o.state[p]["exp_avg"].abs().max())
            torch.testing.assert_close(o.state[q]["exp_avg"], ref_o.state[p]["exp_avg"], rtol=2e-5, atol=1e-6 * scale)
            torch.testing.assert_close(o.state[q]["exp_avg_sq"], ref_o.state[p]["exp_avg_sq"], rtol=2e-5, atol=1e-12)
            saved = o.state_dict()["state"]
        assert all(torch.equal(saved[i]["exp_avg"], o.state[q]["exp_avg"]) for i, q in enumerate(ps))
