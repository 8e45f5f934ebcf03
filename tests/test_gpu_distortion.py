"""The distortion layer of train_step on the device (Trainer.distortion_layer, /root/reference/nerf/utils_wtmk_disen.py:551-577,594) against
the oracle's restatement with the random draw passed in: the layer alone, fused into the decoder's first layer (values + decoder-input
gradient), through train_step (losses + gradients), and inside the captured loop with draws generated on the device."""
import copy
import math
import os

import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr

pytestmark = pytest.mark.gpu
KINDS = ["noise", "brightness", "blurring"]          # applied inside the decoder's first launch
GEOMETRIC = ["rotation", "scaling"]                  # a resampling launch of their own in front of the decoder


def _layer(kind, shape, seed=3, sigma=None):
    """A DistortionLayer with known draws in its device buffers, and the same draw for the oracle."""
    from nerf_signature_amd.distortion import DistortionLayer
    layer = DistortionLayer(kind, seed)
    layer.draw(shape, torch.device("cuda"))
    if sigma is not None and kind == "blurring":      # (a small drawn sigma is the identity to fp32: pick one that blurs)
        layer.param.fill_(sigma)
    if kind == "rotation":        # the oracle takes torchvision's parameter, degrees per image
        draw = [math.degrees(math.atan2(float(sn), float(c))) for c, sn in layer.param.cpu().reshape(-1, 2)]
    elif kind == "scaling":
        draw = layer.factor
    else:
        draw = layer.noise.cpu().clone() if kind == "noise" else float(layer.param.cpu())
    return layer, draw


@pytest.mark.parametrize("kind", GEOMETRIC)
@pytest.mark.parametrize("shape", [(32, 12, 12, 3), (5, 9, 14, 3), (3, 40, 33, 3), (2, 1, 7, 3)])
def test_geometric_layers_alone_match_the_stock_operators_forward_and_backward(kind, shape):
    """wm_distort_geom_fwd / _bwd (rotation: nearest-neighbour resampling about the centre; scaling: 1-d linear interpolation along W to floor(W * sf)
    columns) against the oracle's grid_sample / F.interpolate statement and its autograd: values outside [0, 1] so that the clamp and its mask matter,
    non-square images, several draws.  rotation moves pixels: bit-exact values."""
    from nerf_signature_amd.distortion import _DistortGeometry, reference_ops, scaled_width
    rng = np.random.RandomState(0)
    raw = torch.from_numpy(rng.uniform(-0.25, 1.25, shape).astype(np.float32))
    for seed in (3, 4, 5):
        layer, draw = _layer(kind, shape, seed=seed)
        Wo = layer.out_width(shape[2])
        r = torch.from_numpy(rng.randn(shape[0], shape[1], Wo, 3).astype(np.float32))
        a = raw.cuda().requires_grad_(True)
        out1, clamped = _DistortGeometry.apply(a, layer.kind, layer.param, Wo)
        (out1 * r.cuda()).sum().backward()
        b = raw.clone().requires_grad_(True)
        out0 = fr.distortion_layer(torch.clamp(b, 0, 1), kind, draw)
        (out0 * r).sum().backward()
        assert torch.equal(clamped.cpu(), raw.clamp(0, 1)) and tuple(out1.shape) == tuple(out0.shape)
        if kind == "rotation":
            assert torch.equal(out1.detach().cpu(), out0.detach())
            assert torch.equal(out1.detach().cpu(), reference_ops(raw.clamp(0, 1), kind, layer.param.cpu()))      # the product's host-side statement: same pixels
        else:
            assert Wo == scaled_width(shape[2], draw)
            np.testing.assert_allclose(out1.detach().cpu().numpy(), out0.detach().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=5e-6)
        assert float(b.grad.abs().max()) > 0
    if kind == "scaling":         # factor 1 is the identity; the columns of a constant image stay constant
        layer.set_scaling(1.0)
        ident = _DistortGeometry.apply(raw.cuda(), layer.kind, layer.param, shape[2])[0]
        np.testing.assert_allclose(ident.cpu().numpy(), raw.clamp(0, 1).numpy(), atol=1e-6)
    else:                         # angle 0 is the identity
        layer.set_rotation([0.0] * shape[0])
        assert torch.equal(_DistortGeometry.apply(raw.cuda(), layer.kind, layer.param, shape[2])[0].cpu(), raw.clamp(0, 1))


@pytest.mark.parametrize("kind", KINDS)
def test_layer_alone_matches_the_oracle_forward_and_backward(kind):
    from nerf_signature_amd.distortion import _Distort
    rng = np.random.RandomState(0)
    raw = torch.from_numpy(rng.uniform(-0.25, 1.25, (32, 12, 12, 3)).astype(np.float32))      # values outside [0, 1]: the clamp and its mask matter
    r = torch.from_numpy(rng.randn(32, 12, 12, 3).astype(np.float32))
    layer, draw = _layer(kind, tuple(raw.shape), sigma=0.45)
    a = raw.cuda().requires_grad_(True)
    out1 = _Distort.apply(a, layer.kind, layer.param, layer.noise)
    (out1 * r.cuda()).sum().backward()
    b = raw.clone().requires_grad_(True)
    out0 = fr.distortion_layer(torch.clamp(b, 0, 1), kind, draw)
    (out0 * r).sum().backward()
    np.testing.assert_allclose(out1.detach().cpu().numpy(), out0.detach().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=0, atol=5e-6)
    if kind == "brightness":      # both regimes present: f * x clipped at 1 somewhere, and not clipped somewhere
        f = draw
        x = raw.clamp(0, 1)
        assert bool(((f * x) > 1).any()) or f <= 1.0
    if kind == "blurring":        # sigma -> 0 is the identity, and the taps are normalised (a constant image stays constant)
        layer.param.fill_(0.01)
        ident = _Distort.apply(raw.cuda(), layer.kind, layer.param, None)
        np.testing.assert_allclose(ident.cpu().numpy(), raw.clamp(0, 1).numpy(), atol=1e-6)
        layer.param.fill_(0.5)
        const = _Distort.apply(torch.full((2, 5, 7, 3), 0.37, device="cuda"), layer.kind, layer.param, None)
        np.testing.assert_allclose(const.cpu().numpy(), 0.37, atol=1e-6)


@pytest.mark.parametrize("kind", KINDS + GEOMETRIC)
@pytest.mark.parametrize("shape", [(32, 12, 12, 3), (48, 11, 15, 3), (4, 2, 2, 3)])
def test_fused_into_the_decoder_values_and_input_gradient(kind, shape):
    """decode_rendered(image, layer) == decoder(normalize(distortion_layer(clamp(image)))) of the oracle: logits, the clamped image reported as
    pred_rgb (undistorted, :592), the gradient with respect to the rendered image (what flows back into the block render) and the decoder's
    parameter gradients."""
    from nerf_signature_amd.hidden_models import get_hidden_decoder_multi_views
    torch.manual_seed(5)
    dec = get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=3, channels=64).cuda()
    dec0 = copy.deepcopy(dec).cpu()
    rng = np.random.RandomState(1)
    raw = torch.from_numpy(rng.uniform(-0.2, 1.2, shape).astype(np.float32))
    msg = torch.from_numpy(rng.randint(0, 2, (shape[0], 1)).astype(np.float32))
    layer, draw = _layer(kind, shape, sigma=0.45)
    a = raw.cuda().requires_grad_(True)
    decoded1, pred1 = dec.decode_rendered(a, layer)
    loss1 = torch.nn.functional.binary_cross_entropy_with_logits(decoded1 * 10.0, msg.cuda())
    loss1.backward()
    b = raw.clone().requires_grad_(True)
    pred0 = torch.clamp(b, 0, 1)
    decoded0 = dec0(fr.normalize_img(fr.distortion_layer(pred0, kind, draw).permute(0, 3, 1, 2)))
    loss0 = torch.nn.functional.binary_cross_entropy_with_logits(decoded0 * 10.0, msg)
    loss0.backward()
    assert torch.equal(pred1.cpu(), pred0.detach())
    np.testing.assert_allclose(decoded1.detach().cpu().numpy(), decoded0.detach().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(float(loss1.detach()), float(loss0.detach()), rtol=1e-3, atol=1e-4)
    g1, g0 = a.grad.cpu(), b.grad
    assert float(g0.abs().max()) > 0
    assert float((g1 - g0).norm() / g0.norm()) < 1e-3                                  # the decoder-input gradient
    assert float((g1 - g0).abs().max()) < 1e-3 * float(g0.abs().max()) + 1e-7
    assert torch.equal(g1 == 0, g0 == 0) or float(((g1 == 0) != (g0 == 0)).float().mean()) < 1e-3      # the clamp masks agree
    d1 = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in dec.parameters()]).cpu()
    d0 = torch.cat([p.grad.reshape(-1) for p in dec0.parameters()])
    assert float((d1 - d0).norm() / d0.norm()) < 2e-2


@pytest.mark.parametrize("kind", KINDS + GEOMETRIC)
def test_train_step_with_distortion_vs_oracle(kind, strict_mlp):
    import test_gpu_render as T
    from nerf_signature_amd import trainer
    m, bitfield, C = T._model()
    P, S = T._oracle_params(m, bitfield, C)
    bo, bd, co, cd, gt = T._data()
    msg = torch.from_numpy(cf.messages(32)[2])
    dec_cpu = copy.deepcopy(m.msg_decoder).cpu()
    layer, draw = _layer(kind, tuple(bo.shape), sigma=0.45)
    ref = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec_cpu, distortion=kind, draw=draw, dt_gamma=0.0, max_steps=1024)
    ref["loss"].backward()
    plain = fr.train_step(bo, bd, co, cd, gt, msg, P, S, copy.deepcopy(dec_cpu), dt_gamma=0.0, max_steps=1024)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    pred_rgb, _, content_rgb, lossi, lossw, loss = trainer.train_step(m, data, msg, dict(dt_gamma=0, max_steps=1024), distortion=layer)
    loss.backward()
    assert abs(float(ref["lossw"]) - float(plain["lossw"])) > 1e-4                     # the layer changes what the decoder sees
    np.testing.assert_allclose(pred_rgb.detach().cpu().numpy(), ref["pred_rgb"].detach().numpy(), rtol=0, atol=1e-3)      # pred_rgb stays undistorted
    np.testing.assert_allclose(float(lossw.detach()), float(ref["lossw"].detach()), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(float(lossi.detach()), float(ref["lossi"].detach()), rtol=1e-3)
    np.testing.assert_allclose(float(loss.detach()), float(ref["loss"].detach()), rtol=1e-3, atol=1e-3)
    bits = [int(v) for v in msg]
    G0, G1 = P["cb_tables"][bits[0]].grad, m.msg_encoder.embeddings[bits[0]].weight.grad.cpu()
    assert float(G0.abs().max()) > 0 and float((G1 - G0).norm() / G0.norm()) < 5e-3
    d1 = T._decoder_grad_vector(m.msg_decoder).cpu()
    d0 = torch.cat([p.grad.reshape(-1) for p in dec_cpu.parameters()])
    assert float((d1 - d0).norm() / d0.norm()) < 2e-2


def test_device_draws_are_a_function_of_seed_and_step_with_the_right_distributions():
    from nerf_signature_amd.distortion import DistortionLayer
    dev = torch.device("cuda")
    shape = (32, 12, 12, 3)
    step = torch.zeros(1, dtype=torch.int32, device=dev)

    def draws(kind, seed, k):
        layer = DistortionLayer(kind, seed)
        step.fill_(k)
        layer.draw_on_device(step, shape, dev)
        return layer.noise.cpu().clone() if kind == "noise" else float(layer.param.cpu())

    n = torch.cat([draws("noise", 7, k).reshape(-1) for k in range(40)])               # 553 k samples
    assert abs(float(n.mean())) < 2e-3 and abs(float(n.var()) - 0.1) < 2e-3            # N(0, 0.1): utils_wtmk_disen.py:555
    assert abs(float((n ** 4).mean()) / float(n.var()) ** 2 - 3.0) < 0.05              # Gaussian kurtosis
    assert float(torch.corrcoef(torch.stack([n[:-1], n[1:]]))[0, 1].abs()) < 5e-3       # neighbouring elements uncorrelated
    assert torch.equal(draws("noise", 7, 3), draws("noise", 7, 3))                     # a pure function of (seed, step)
    assert not torch.equal(draws("noise", 7, 3), draws("noise", 7, 4)) and not torch.equal(draws("noise", 7, 3), draws("noise", 8, 3))
    f = np.array([draws("brightness", 1, k) for k in range(400)])
    s = np.array([draws("blurring", 1, k) for k in range(400)])
    assert 0.5 <= f.min() < 0.52 and 1.48 < f.max() <= 1.5 and abs(f.mean() - 1.0) < 0.05       # ColorJitter(brightness=0.5): U[0.5, 1.5]
    assert 0.01 <= s.min() < 0.02 and 0.48 < s.max() <= 0.5 and abs(s.mean() - 0.255) < 0.03    # GaussianBlur sigma: U[0.01, 0.5]
    rot = DistortionLayer("rotation", 1)
    angles = []
    for k in range(40):
        step.fill_(k)
        rot.draw_on_device(step, shape, dev)
        c, sn = rot.param.cpu().reshape(-1, 2).unbind(1)
        assert rot.param.numel() == 2 * shape[0] and torch.allclose(c * c + sn * sn, torch.ones(shape[0]), atol=1e-6)
        angles += [math.degrees(math.atan2(float(b), float(a))) for a, b in zip(c, sn)]
    angles = np.array(angles)                                                           # RandomRotation((-30, 30)): one angle per image, U[-30, 30]
    assert -30.0 <= angles.min() < -29.5 and 29.5 < angles.max() <= 30.0 and abs(angles.mean()) < 1.5 and abs(angles.std() - 60 / math.sqrt(12)) < 0.8
    assert len(set(np.round(angles[:32], 4))) == 32                                     # per image, not per call
    with pytest.raises(NotImplementedError):
        DistortionLayer("scaling").draw_on_device(step, shape, dev)                    # the factor decides a shape: a host value


@pytest.mark.parametrize("kind", KINDS + GEOMETRIC)
def test_captured_loop_draws_on_the_device_and_matches_the_eager_step(kind):
    """GraphedWatermarkLoop(distortion=...): every replay re-draws the layer's parameters on the device and the decoder's first launch applies them.
    With a learning rate of zero the parameters stay put, so an eager train_step on the same message with the buffers the last replay left behind
    must reproduce that replay's losses; and consecutive replays must have used different draws."""
    import test_gpu_render as T
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam
    bo, bd, co, cd, gt = T._data(n_content=300)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    torch.manual_seed(0)
    m, _, _ = T._model()
    opt = CodebookAdam(m.get_params(0.0), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    kw = dict(dt_gamma=0, max_steps=1024)
    loop = trainer.GraphedWatermarkLoop(m, opt, kw, data, distortion=kind, distortion_seed=11)
    msg = torch.from_numpy(np.random.RandomState(3).randint(0, 2, 32).astype(np.float32))
    seen, losses, widths = [], [], set()
    for _ in range(3 if kind != "scaling" else 12):
        out = loop.step(msg)
        torch.cuda.synchronize()
        losses.append(float(out[4].detach()))
        seen.append(loop.distortion.noise.cpu().clone() if kind == "noise" else loop.distortion.param.cpu().clone())
        widths.add(loop.distortion.out_width(bo.shape[2]))
        assert not loop.overflowed()
    assert all(not torch.equal(seen[i], seen[i + 1]) for i in range(len(seen) - 1))
    # same message, same weights, different distortion: different loss (scaling: steps whose width equals W see the undistorted blocks -- the operator's copy case)
    assert len(set(round(l, 6) for l in losses)) >= (len(losses) if kind != "scaling" else len(widths))
    if kind == "scaling":         # one capture per decoder input width; the host-side draws of 12 steps visit several of them
        W = bo.shape[2]
        assert sorted(loop.variants) == list(range(int(0.75 * W), int(1.25 * W - 1e-9) + 1)) and len(widths) >= 3 and widths <= set(loop.variants)
    layer = loop.distortion
    loop.close()
    eager = trainer.train_step(m, data, msg, kw, distortion=layer)                      # the buffers still hold the last replay's draws
    np.testing.assert_allclose(float(eager[4].detach()), losses[-1], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", GEOMETRIC)
def test_geometric_distortions_run_in_the_eager_loop(kind):
    """rotation / scaling (utils_wtmk_disen.py:557-566) in the eager loop with the loop's own per-step draws: one step trains through them (scaling hands
    the decoder a different width), gradients reach the codebook."""
    import test_gpu_render as T
    from nerf_signature_amd import trainer
    bo, bd, co, cd, gt = T._data(n_content=200)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    m, _, _ = T._model()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = trainer.WatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024), distortion=kind)
    msg = torch.from_numpy(np.random.RandomState(3).randint(0, 2, 32).astype(np.float32))
    before = [e.weight.detach().clone() for e in m.msg_encoder.embeddings]
    out = loop.step(data, msg)
    assert np.isfinite(float(out[5].detach()))
    moved = sum(int(not torch.equal(a, e.weight.detach())) for a, e in zip(before, m.msg_encoder.embeddings))
    assert moved == 32
