"""GPU parity of the model-level path (NeRFNetwork.render / train_step) against values produced by the
reference's own renderer glue (golden G9) and against the CPU oracle.  Tolerance of the path: 1e-3 on RGB /
sigma / losses (north_star); integer ray records bit-exact."""
import os

import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(D=32, bound=1.0):
    from nerf_signature_amd.network import NeRFNetwork
    m = NeRFNetwork(bound=bound, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
    grid, bitfield, C = cf.ball_scene(bound=bound)
    with torch.no_grad():
        for l in range(16):
            m.encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(l)))
        for l in range(2 * D):
            m.msg_encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(100 + l, scale=0.05)))
        m.sigma_net.params.copy_(torch.from_numpy(cf.mlp_params(3072, 1337)))
        m.color_net.params.copy_(torch.from_numpy(cf.mlp_params(7168, 1338)))
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bitfield))
    return m.cuda().train(), bitfield, C


def _oracle_params(m, bitfield, C, bound=1.0):
    P = {"bound": bound, "base_tables": [e.weight.detach().cpu() for e in m.encoder.embeddings],
         "cb_tables": [e.weight.detach().cpu().clone().requires_grad_(True) for e in m.msg_encoder.embeddings],
         "sigma_params": m.sigma_net.params.detach().cpu(), "color_params": m.color_net.params.detach().cpu()}
    S = {"bound": bound, "cascade": C, "grid_size": 128, "density_bitfield": bitfield, "aabb": np.array([-bound] * 3 + [bound] * 3, np.float32),
         "min_near": 0.2, "density_scale": 1}
    return P, S


def test_render_matches_reference_glue_golden(mlp_prec):
    g = np.load(os.path.join(G, "g8_g9_glue.npz"))
    m, _, _ = _model()
    o, d, msg = (torch.from_numpy(g[k]).cuda() for k in ("rays_o", "rays_d", "msg"))
    opt_namespace = dict(dt_gamma=0, max_steps=1024, some_unrelated_flag=123, lr=1e-2, workspace="x", num_rays=4096)  # vars(opt) is splatted in
    out = m.render(o, d, msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **opt_namespace)
    assert set(out) == {"image", "depth", "weights_sum"} and out["image"].shape == (1, 64, 3) and out["depth"].shape == (1, 64)
    np.testing.assert_allclose(out["image"].detach().cpu().numpy(), g["image"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(out["weights_sum"].detach().cpu().numpy(), g["weights_sum"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(out["depth"].detach().cpu().numpy(), g["depth"], rtol=0, atol=1e-3, equal_nan=True)
    assert float(np.abs(out["image"].detach().cpu().numpy() - g["image"]).max()) < {"bf16x3": 5e-5, "f16": 3e-4}[mlp_prec]
    (out["image"] * torch.from_numpy(g["gvec"]).cuda()).sum().backward()
    bits = [int(v) for v in g["msg"]]
    g0 = m.msg_encoder.embeddings[bits[0]].weight.grad
    assert m.msg_encoder.embeddings[1 - bits[0]].weight.grad is None      # unselected tables: grad None, as in the reference
    assert all(p.grad is None for p in m.encoder.parameters()) and m.sigma_net.params.grad is None
    nz = torch.nonzero(g0.abs().sum(-1)).squeeze(-1)
    np.testing.assert_array_equal(nz.cpu().numpy().astype(np.int32), g["cb_grad_rows"])
    scale = float(np.abs(g["cb_grad_vals"]).max())
    if mlp_prec == "bf16x3":
        np.testing.assert_allclose(g0[nz].cpu().numpy(), g["cb_grad_vals"], rtol=1e-3, atol=1e-4 * scale)
    else:       # fp16 operands: ReLU kinks taken on the other side at a few per cent of the points (conftest.strict_mlp) -- aggregate bound
        rel = float(np.linalg.norm(g0[nz].cpu().numpy() - g["cb_grad_vals"]) / np.linalg.norm(g["cb_grad_vals"]))
        assert rel < 5e-2, rel
    with torch.no_grad():
        st = m.render(o, d, msg, staged=True, max_ray_batch=24, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
        cl = m.render(o, d, None, staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
        m.eval()
        ev = m.render(o, d, msg, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    assert set(st) == {"image", "depth"}
    np.testing.assert_allclose(st["image"].cpu().numpy(), g["image_staged"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(cl["image"].cpu().numpy(), g["image_clean"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(ev["image"].cpu().numpy(), g["image_eval"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(ev["depth"].cpu().numpy(), g["depth_eval"], rtol=0, atol=1e-3, equal_nan=True)


def _data(n_content=512, seed=0, scene_bound=1.0, block=6, shift=0):
    pose, intr, inds = cf.orbit_rays(n_content, seed=seed)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    # D blocks of block x block pixels around the image centre (rays through the ball)
    full_o, full_d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400)
    full_o, full_d = full_o.view(400, 400, 3), full_d.view(400, 400, 3)
    bo, bd = [], []
    for k in range(32):
        r, c = 150 + shift + (k // 8) * 20, 130 + shift + (k % 8) * 18
        bo.append(full_o[r:r + block, c:c + block])
        bd.append(full_d[r:r + block, c:c + block])
    gt = torch.from_numpy(np.random.RandomState(5).rand(1, n_content, 3).astype(np.float32))
    return torch.stack(bo).contiguous(), torch.stack(bd).contiguous(), o.contiguous(), d.contiguous(), gt


def _decoder_grad_vector(decoder):
    """All decoder gradients as one vector; a conv bias in front of a BatchNorm has an identically zero gradient, which the
    GPU path reports as None."""
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in decoder.parameters()])


def test_train_step_losses_and_gradients_vs_oracle(mlp_prec):
    from nerf_signature_amd import trainer
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    bo, bd, co, cd, gt = _data()
    msg = torch.from_numpy(cf.messages(32)[2])
    import copy
    dec_cpu = copy.deepcopy(m.msg_decoder).cpu()
    ref = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec_cpu, dt_gamma=0.0, max_steps=1024)
    ref["loss"].backward()
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    pred_rgb, gt_rgb, content_rgb, lossi, lossw, loss = trainer.train_step(m, data, msg, dict(dt_gamma=0, max_steps=1024, junk=1))
    loss.backward()
    # integer parity: the same rays produce the same number of points
    assert int(m.step_counter[0, 0]) == ref["block"]["n_points"] and int(m.step_counter[1, 0]) == ref["content"]["n_points"]
    np.testing.assert_allclose(pred_rgb.detach().cpu().numpy(), ref["pred_rgb"].detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(content_rgb.detach().cpu().numpy(), ref["content_pred_rgb"].detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(float(lossi.detach()), float(ref["lossi"].detach()), rtol=1e-3)
    np.testing.assert_allclose(float(lossw.detach()), float(ref["lossw"].detach()), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(float(loss.detach()), float(ref["loss"].detach()), rtol=1e-3, atol=1e-3)
    bits = [int(v) for v in msg]
    G0 = P["cb_tables"][bits[0]].grad
    G1 = m.msg_encoder.embeddings[bits[0]].weight.grad
    scale = float(G0.abs().max())
    assert scale > 0
    # A ReLU whose pre-activation is within rounding of zero can fall on different sides in the fp32 oracle and in
    # the split-bf16 kernel; its point then contributes a different (finite) gradient to 16 table entries.  About one
    # in 1e5 activations does, so the gradient is compared in the aggregate plus an outlier budget, not element-wise.
    diff = (G1.cpu() - G0)
    rel_G = float(diff.norm() / G0.norm())
    print(f"\n[{mlp_prec}] shared codebook gradient rel. L2 vs oracle: {rel_G:.3e}")
    # fp16 operands: a few per cent of the points take a ReLU kink on the other side (test_field_backward_away_from_relu_kinks)
    assert rel_G < {"bf16x3": 5e-3, "f16": 5e-2}[mlp_prec]
    assert float((diff.abs() > 2e-3 * scale).float().mean()) < {"bf16x3": 1e-3, "f16": 2e-2}[mlp_prec]
    if mlp_prec == "f16":
        # ... and the default arithmetic against the oracle that rounds the matrix operands where the kernels do (fr.round_operand): the same side of
        # every kink, so the whole step's codebook gradient agrees in the split-bf16 class (round 4: the default mode pinned, not only bounded)
        P16, _ = _oracle_params(m, bitfield, C)
        with torch.no_grad():
            for t_new, t_old in zip(P16["cb_tables"], P["cb_tables"]):
                t_new.copy_(t_old)
        P16["mlp_operands"] = "f16"
        ref16 = fr.train_step(bo, bd, co, cd, gt, msg, P16, S, copy.deepcopy(dec_cpu), dt_gamma=0.0, max_steps=1024)
        ref16["loss"].backward()
        G16 = P16["cb_tables"][bits[0]].grad
        rel16 = float((G1.cpu() - G16).norm() / G16.norm())
        print(f"[f16] shared codebook gradient rel. L2 vs the operand-rounding oracle: {rel16:.3e}")
        assert rel16 < 5e-3 and float(((G1.cpu() - G16).abs() > 2e-3 * scale).float().mean()) < 1e-3
        np.testing.assert_allclose(float(lossw.detach()), float(ref16["lossw"].detach()), rtol=2e-4, atol=2e-4)
    # decoder gradients as one vector (conv biases in front of a BatchNorm have a mathematically zero gradient,
    # so a per-tensor relative comparison would compare rounding noise)
    d1 = _decoder_grad_vector(m.msg_decoder).cpu()
    d0 = torch.cat([p.grad.reshape(-1) for p in dec_cpu.parameters()])
    assert float((d1 - d0).norm() / d0.norm()) < 2e-2
    # without the conv biases (mathematically zero gradient: pure rounding noise on the oracle side, None on the GPU side)
    keep = [n for n, _ in m.msg_decoder.named_parameters() if not n.endswith("layers.0.bias")]
    p1, p0 = dict(m.msg_decoder.named_parameters()), dict(dec_cpu.named_parameters())
    assert len(keep) == len(p1) - 9
    v1 = torch.cat([p1[n].grad.reshape(-1).cpu() for n in keep])
    v0 = torch.cat([p0[n].grad.reshape(-1) for n in keep])
    rel = float((v1 - v0).norm() / v0.norm())
    print(f"\n[{mlp_prec}] decoder gradient (no conv biases) rel. L2 vs oracle: {rel:.3e}")
    assert rel < {"bf16x3": 1e-3, "f16": 5e-3}[mlp_prec]


def test_loop_step_with_sink_equals_autograd_path():
    """The GradSink route (shared gradient, fan-out) produces the same parameter update as plain autograd."""
    from nerf_signature_amd import trainer
    bo, bd, co, cd, gt = _data(n_content=256)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    msg = torch.from_numpy(cf.messages(32)[2])
    grads = []
    for use_sink in (False, True):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        loop = trainer.WatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024), use_sink=use_sink)
        sel, uns = m.msg_encoder.embeddings[int(msg[0])].weight, m.msg_encoder.embeddings[1 - int(msg[0])].weight
        before, untouched = sel.detach().clone(), uns.detach().clone()
        out = loop.step(data, msg)
        assert not torch.equal(before, sel.detach())
        assert torch.equal(untouched, uns.detach()) and uns.grad is None      # Adam skips grad=None tables
        # every selected table carries the same gradient
        g = sel.grad.clone()
        other = m.msg_encoder.embeddings[2 * 5 + int(msg[5])].weight.grad
        assert torch.equal(g, other)
        grads.append((g, float(out[-1].detach()), [_decoder_grad_vector(m.msg_decoder)]))
    assert abs(grads[0][1] - grads[1][1]) < 1e-5
    # float atomics make the two routes differ in the last bits only
    assert float((grads[0][0] - grads[1][0]).norm() / grads[0][0].norm()) < 1e-5
    a, b = (torch.cat([t.reshape(-1) for t in g[2]]) for g in grads)
    assert float((a - b).norm() / a.norm()) < 1e-3


def test_codebook_adam_matches_torch_adam():
    """opt_codebook_adam == torch.optim.Adam over several steps, including a table that is skipped for a step."""
    from nerf_signature_amd.optim import CodebookAdam
    torch.manual_seed(0)
    T = 1 << 19
    init = [torch.randn(T, 2, device="cuda") * 0.05 for _ in range(3)]
    ref_p = [torch.nn.Parameter(t.clone()) for t in init]
    our_p = [torch.nn.Parameter(t.clone()) for t in init]
    extra_ref, extra_our = torch.nn.Parameter(torch.ones(7, device="cuda")), torch.nn.Parameter(torch.ones(7, device="cuda"))
    kw = dict(betas=(0.9, 0.99), eps=1e-15)
    ref = torch.optim.Adam([{"params": ref_p, "lr": 1e-2}, {"params": [extra_ref], "lr": 1e-2}], **kw)
    our = CodebookAdam([{"params": our_p, "lr": 1e-2}, {"params": [extra_our], "lr": 1e-2}], **kw)
    v0 = our_p[0]._version
    for it in range(5):
        G = torch.randn(T, 2, device="cuda") * (10.0 ** -it)
        G[::3] = 0                                     # untouched rows: zero gradient
        sel = [0, 1, 2] if it != 2 else [0, 2]         # table 1 is not selected at step 2
        ref.zero_grad(set_to_none=True)
        for i in sel:
            ref_p[i].grad = G.clone()
        extra_ref.grad = torch.full_like(extra_ref, 0.5)
        extra_our.grad = torch.full_like(extra_our, 0.5)
        ref.step()
        our.step_shared([our_p[i] for i in sel], G)
        our.step()
    for a, b in zip(ref_p, our_p):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=1e-7)
    assert torch.equal(extra_ref, extra_our)
    assert float(our.state[our_p[1]]["step"]) == 4.0 and float(our.state[our_p[0]]["step"]) == 5.0
    np.testing.assert_allclose(our.state[our_p[1]]["exp_avg_sq"].cpu().numpy(), ref.state[ref_p[1]]["exp_avg_sq"].cpu().numpy(), rtol=2e-5, atol=1e-12)
    assert our_p[0]._version > v0                      # caches keyed on tensor versions see the native update
    sd = our.state_dict()                              # torch-format state: loads into a plain Adam
    ref.load_state_dict(sd)


def test_codebook_adam_sel_with_next_presum():
    """opt_codebook_adam_sel_next: the same parameter / state update as opt_codebook_adam_sel, and S_next is bit-identical to the
    pre-sum of the NEXT message over the tables as updated (odd D exercises the tail of the two-tables-per-trip loop)."""
    from nerf_signature_amd import fieldops as fo
    from nerf_signature_amd.optim import CodebookAdam
    for D in (5, 32):
        torch.manual_seed(D)
        init = [torch.randn(1 << 19, 2, device="cuda") * 0.05 for _ in range(2 * D)]
        runs = []
        for with_next in (False, True):
            ps = [torch.nn.Parameter(t.clone()) for t in init]
            opt = CodebookAdam([{"params": ps, "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15, capturable=True)
            lr = torch.tensor(1e-2, device="cuda")
            S = torch.empty(1 << 19, 2, device="cuda")
            for it in range(3):
                torch.manual_seed(100 * D + it)
                G = torch.randn(1 << 19, 2, device="cuda") * (10.0 ** -it)
                msg = torch.from_numpy(np.random.RandomState(10 * D + it).randint(0, 2, D).astype(np.float32)).cuda()
                nxt = torch.from_numpy(np.random.RandomState(10 * D + it + 1).randint(0, 2, D).astype(np.float32)).cuda()
                if with_next:
                    opt.step_shared_sel(ps, msg, G, lr, 0.5, next_message_dev=nxt, S_next=S)
                    assert torch.equal(S, fo.codebook_presum_sel(ps, nxt))
                else:
                    opt.step_shared_sel(ps, msg, G, lr, 0.5)
            runs.append(([p.detach().clone() for p in ps], [float(opt.state[p]["step"]) if len(opt.state[p]) else 0.0 for p in ps]))
        assert runs[0][1] == runs[1][1] and sum(runs[0][1]) == 3 * D
        assert all(torch.equal(a, b) for a, b in zip(runs[0][0], runs[1][0]))


def test_graphed_loop_matches_eager_loop():
    """GraphedWatermarkLoop (capacity march, device-side table selection, captured hipGraph) == WatermarkLoop step for step."""
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam
    bo, bd, co, cd, gt = _data(n_content=300)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    msgs = [torch.from_numpy(np.random.RandomState(s).randint(0, 2, 32).astype(np.float32)) for s in range(4)]
    lr_lambda = lambda it: 0.5 ** it
    runs = []
    for graphed in (False, True, "announced", "half-announced"):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=bool(graphed))
        if graphed:
            loop = trainer.GraphedWatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024), data, lr_lambda=lr_lambda)
            # no host synchronisation between the steps (how the loop is meant to run): the host is then several replays ahead of
            # the GPU, which is what once let a step read a later step's message out of the pinned staging buffer.
            # "announced": every next message handed over one step early (the optimiser kernel leaves its pre-sum behind);
            # "half-announced": only some, and one of them wrongly -- those steps must fall back to the stand-alone pre-sum
            nxt = {True: lambda k: None, "announced": lambda k: msgs[k + 1] if k + 1 < len(msgs) else None,
                   "half-announced": lambda k: {0: msgs[1], 1: 1 - msgs[2]}.get(k)}[graphed]
            held = [loop.step(msg, next_message=nxt(k))[5].detach().clone() for k, msg in enumerate(msgs)]
            losses = [float(v) for v in held]
            assert not loop.overflowed()
        else:
            sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)
            loop = trainer.WatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024), lr_scheduler=sched)
            losses = [float(loop.step(data, msg)[5].detach()) for msg in msgs]
        torch.cuda.synchronize()
        tables = [e.weight.detach().clone() for e in m.msg_encoder.embeddings]
        dec = torch.cat([p.detach().reshape(-1) for p in m.msg_decoder.parameters()])
        steps = [float(opt.state[e.weight]["step"]) if len(opt.state[e.weight]) else 0.0 for e in m.msg_encoder.embeddings]
        runs.append((losses, tables, dec, steps))
    for other in runs[2:]:      # the look-ahead variants replay the same graph: identical to the plain graphed run up to atomics' order
        np.testing.assert_allclose(other[0], runs[1][0], rtol=1e-4, atol=1e-5)
        assert other[3] == runs[1][3]
    (l0, t0, d0, s0), (l1, t1, d1, s1) = runs[:2]
    np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=2e-4)
    assert s0 == s1 and sum(s0) == 4 * 32                       # per-table step counts: one per selection
    # Adam normalises every touched row to a step of ~lr, so rows whose tiny gradient differs in the last bits (float
    # atomics) can move differently; compare in aggregate
    num = sum(float((a - b).pow(2).sum()) for a, b in zip(t0, t1)) ** 0.5
    den = sum(float((a - cf_t).pow(2).sum()) for a, cf_t in zip(t0, [torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda() for l in range(64)])) ** 0.5
    assert den > 0 and num / den < 0.05
    # (the decoder's conv biases sit in front of BatchNorms: their gradient is rounding noise, and Adam with eps=1e-15 turns
    #  noise into +-lr steps, so the parameter vectors agree only to a few per cent while the loss trajectories agree to 2e-3)
    assert float((d0 - d1).norm() / d0.norm()) < 0.05


def test_split_graphs_with_rccl_exchange_rehearsed_on_one_rank():
    """The multi-rank execution -- graph(forward+backward) | RCCL all-reduces of G and the decoder's flat gradient buffer |
    graph(optimiser, next march) -- rehearsed on this one GPU through a world-size-1 "nccl" process group
    (NERFSIG_FORCE_EXCHANGE=1): sums over one rank change nothing, so the losses must equal the single-graph loop's."""
    import torch.distributed as dist
    from nerf_signature_amd import dp, trainer
    from nerf_signature_amd.optim import CodebookAdam
    bo, bd, co, cd, gt = _data(n_content=300)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    msgs = [torch.from_numpy(np.random.RandomState(s).randint(0, 2, 32).astype(np.float32)) for s in range(4)]

    def run(**loop_kw):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
        loop = trainer.GraphedWatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024), data, **loop_kw)
        held = [loop.step(msg)[5].detach().clone() for msg in msgs]
        torch.cuda.synchronize()
        return loop, [float(v) for v in held], torch.cat([e.weight.detach().reshape(-1) for e in m.msg_encoder.embeddings])

    loop0, l0, t0 = run()
    assert len(loop0.segments) == 1 and loop0.exchange.bytes_per_step == 0
    assert not dist.is_initialized()
    os.environ["NERFSIG_FORCE_EXCHANGE"] = "1"
    try:
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{29700 + os.getpid() % 200}", rank=0, world_size=1)
        assert dp.exchange_active()
        loop1, l1, t1 = run()
        # render | all-gather | decode+backward | all-reduce | optimiser (own tables) | all-reduce of the partial pre-sums (closes the step)
        assert len(loop1.segments) == 3 and len(loop1.between) == 3 and loop1.sharded and loop1.opt_shard == (0, 32)
        assert loop1.exchange.collectives_per_step == 1                  # G and the decoder's gradient block travel together
        assert loop1.exchange.bytes_per_step == (1 << 19) * 2 * 4 + sum(p.numel() for p in loop1.model.msg_decoder.parameters() if p.grad is not None) * 4
        # sharded blocks take the "beside" schedule: block render captured first, the content render's backward issued first from its own stream,
        # the decoder's parameter gradients on a third stream
        assert loop1.content_backward_first and loop1._blocks_issued_first() and loop1.weights_stream is not loop1.side_stream
        # ... and the same step with the collectives captured INSIDE the graph (what bench.py's launcher tries first for N > 1): one segment, the
        # content render not joined in front of the all-gather
        os.environ["NERFSIG_CAPTURE_COLLECTIVES"] = "1"
        loop2, l2, t2 = run()
        assert len(loop2.segments) == 1 and len(loop2.between) == 0 and loop2.sharded and loop2.content_backward_first
        # segmented capture without a side stream (overlap_content=False: one stream, no early content backward): the all-gather ends a segment and nothing
        # of the new segment may wait for a stream that was never forked into it
        os.environ["NERFSIG_CAPTURE_COLLECTIVES"] = "0"
        loop3, l3, t3 = run(overlap_content=False)
        assert len(loop3.segments) == 3 and loop3.sharded and loop3.side_stream is None and not loop3.content_backward_first
        extra = [l3]
    finally:
        os.environ.pop("NERFSIG_FORCE_EXCHANGE", None)
        os.environ.pop("NERFSIG_CAPTURE_COLLECTIVES", None)
        if dist.is_initialized():
            dist.destroy_process_group()
    np.testing.assert_allclose(l1, l0, rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(l2, l0, rtol=2e-3, atol=2e-4)
    for l3 in extra:
        np.testing.assert_allclose(l3, l0, rtol=2e-3, atol=2e-4)
    assert float((t1 - t0).norm()) <= 0.05 * float((t0 - torch.cat([torch.from_numpy(cf.table(100 + l, scale=0.05)).reshape(-1) for l in range(64)]).cuda()).norm())


def test_graphed_loop_marches_ahead_with_changing_rays():
    """march_ahead: the block render's samples are marched at the end of the previous replay.  With different rays and images
    every step -- handed over one step early (`next_data`) or at the step itself (`data`, which re-marches before the replay) --
    the loss trajectory equals the eager loop's."""
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam

    def make(seed):
        bo, bd, co, cd, gt = _data(n_content=300, seed=seed, shift=3 * seed)
        return {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()},
                "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": (gt * (0.4 + 0.15 * seed)).cuda()}}

    datas = [make(s) for s in range(4)]
    assert not torch.equal(datas[0]["watermark"]["rays_d_block"], datas[1]["watermark"]["rays_d_block"])
    msgs = [torch.from_numpy(np.random.RandomState(10 + s).randint(0, 2, 32).astype(np.float32)) for s in range(4)]
    kw = dict(dt_gamma=0, max_steps=1024)
    runs = {}
    for mode in ("eager", "next_data", "data"):
        torch.manual_seed(0)
        m, _, _ = _model()
        opt = CodebookAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=mode != "eager")
        if mode == "eager":
            loop = trainer.WatermarkLoop(m, opt, kw)
            runs[mode] = [[float(v.detach()) for v in loop.step(datas[k], msgs[k])[3:6]] for k in range(4)]
            continue
        loop = trainer.GraphedWatermarkLoop(m, opt, kw, datas[0], headroom=0.5)
        assert loop.march_ahead
        held = []
        for k in range(4):
            if mode == "next_data":
                out = loop.step(msgs[k], next_data=datas[k + 1] if k + 1 < 4 else None)
            else:
                out = loop.step(msgs[k], data=datas[k])
            held.append([v.detach().clone() for v in out[3:6]])
        runs[mode] = [[float(v) for v in row] for row in held]
        assert not loop.overflowed()
        n_block, n_content = loop.point_counts()
        assert n_block > 0 and n_content > 0
    for mode in ("next_data", "data"):
        np.testing.assert_allclose(runs[mode], runs["eager"], rtol=2e-3, atol=2e-4)
    assert abs(runs["eager"][0][0] - runs["eager"][1][0]) > 1e-4          # the steps really saw different data


def _oracle_and_model(D, bound, scene_seed=0):
    m, bitfield, C = _model(D=D, bound=bound)
    P, S = _oracle_params(m, bitfield, C, bound=bound)
    return m, P, S


def test_counter_like_unbounded_scene_two_cascades():
    """BASELINE config 3 (Mip-NeRF360/counter-like): bound=2 -> two cascades, camera inside the box, dt_gamma=0."""
    m, P, S = _oracle_and_model(32, 2.0)
    assert m.cascade == 2 and m.density_bitfield.numel() == 2 * 128 ** 3 // 8
    pose, intr, inds = cf.orbit_rays(300, seed=4, radius=1.3, focal=555.56 * 0.5)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    msg = torch.from_numpy(cf.messages(32)[2])
    ref = fr.render(o, d, msg, P, S, staged=False, bg_color=1, dt_gamma=0.0, max_steps=1024)
    out = m.render(o.cuda(), d.cuda(), msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
    assert int(m.step_counter[(m.local_step - 1) % 16, 0]) == ref["n_points"] > 0        # integer parity across cascades
    np.testing.assert_allclose(out["image"].detach().cpu().numpy(), ref["image"].detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(out["weights_sum"].detach().cpu().numpy(), ref["weights_sum"].detach().numpy(), rtol=0, atol=1e-3)


def test_fern_like_48bit_staged_full_image_and_decoder():
    """BASELINE config 5 (LLFF/fern-like): bound=2, dt_gamma=1/128, D=48, staged full-image render in max_ray_batch chunks under
    no_grad with the model left in training mode (as test_image does, utils_wtmk_disen.py:832), then the decoder on D blocks."""
    from nerf_signature_amd.trainer import BIT_ACC
    D = 48
    m, P, S = _oracle_and_model(D, 2.0)
    H, W = 30, 44                                    # reduced image (the reference shape is 756x1008: 187 chunks of 4096)
    pose, _, _ = cf.orbit_rays(1, seed=5, radius=1.6)
    intr = np.array([40.0, 40.0, W / 2, H / 2], np.float32)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, H, W)
    msg = torch.from_numpy(cf.messages(D)[2])
    kw = dict(bg_color=1, dt_gamma=1 / 128, max_steps=1024)
    ref = fr.render(o, d, msg, P, S, staged=True, max_ray_batch=256, **kw)
    with torch.no_grad():
        out = m.render(o.cuda(), d.cuda(), msg, staged=True, max_ray_batch=256, perturb=False, force_all_rays=True, **kw)
    np.testing.assert_allclose(out["image"].cpu().numpy(), ref["image"].numpy(), rtol=0, atol=1e-3)
    hit = ~torch.isnan(ref["depth"])
    np.testing.assert_allclose(out["depth"].cpu()[hit].numpy(), ref["depth"][hit].numpy(), rtol=0, atol=1e-3)
    # 48 blocks of 5x5 pixels -> decoder logits -> bit accuracy, same as the oracle's image through the same decoder
    img1 = out["image"].reshape(H, W, 3).clamp(0, 1)
    img0 = ref["image"].reshape(H, W, 3).clamp(0, 1)
    blocks = lambda im: torch.stack([im[(k // 8) * 5:(k // 8) * 5 + 5, (k % 8) * 5:(k % 8) * 5 + 5] for k in range(D)]).permute(0, 3, 1, 2)
    import copy
    with torch.no_grad():
        dec1 = m.msg_decoder(m.normalization(blocks(img1)))
        dec0 = copy.deepcopy(m.msg_decoder).cpu()(fr.normalize_img(blocks(img0)))
    np.testing.assert_allclose(dec1.cpu().numpy(), dec0.numpy(), rtol=0, atol=2e-3)
    a1, a0 = BIT_ACC(), BIT_ACC()
    a1.update(dec1.cpu().permute(1, 0), msg[None])
    a0.update(dec0.permute(1, 0), msg[None])
    assert abs(a1.measure() - a0.measure()) <= 1.0 / D              # within one bit (north_star)


def test_eval_step_and_test_step_mirror_the_reference_callers():
    """trainer.eval_step / test_step (utils_wtmk_disen.py:648-722): the block branch (render, clamp, decode, BCE + MSE) and the whole-view
    branch (staged render against the ground truth) against the oracle's renders and the stock loss arithmetic."""
    import copy
    from nerf_signature_amd import trainer
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    bo, bd, _, _, _ = _data(n_content=8, block=4)
    msg = torch.from_numpy(cf.messages(32)[1])
    kw = dict(dt_gamma=0, max_steps=1024, some_cli_flag=True)
    gt_blocks = torch.rand(32, 4, 4, 3)
    data = {"H": 400, "W": 400, "rays_o_block": bo.cuda(), "rays_d_block": bd.cuda(), "images_block": gt_blocks.cuda()}
    with torch.no_grad():
        pred, depth, gt, decoded, li, lw, l = trainer.eval_step(m, data, msg.cuda(), kw, render_whole=False)
        ref = fr.render(bo, bd, msg, P, S, bg_color=1, dt_gamma=0.0, max_steps=1024)
        img0 = ref["image"].reshape(32, 4, 4, 3).clamp(0, 1)
        dec0 = copy.deepcopy(m.msg_decoder).cpu()(fr.normalize_img(img0.permute(0, 3, 1, 2)))
    np.testing.assert_allclose(pred.cpu().numpy(), img0.numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(decoded.cpu().numpy(), dec0.numpy(), rtol=0, atol=2e-3)
    lw0 = torch.nn.functional.binary_cross_entropy_with_logits(dec0 * 10.0, msg.unsqueeze(-1))
    li0 = ((img0 - gt_blocks) ** 2).mean()
    assert abs(float(lw) - float(lw0)) < 2e-3 and abs(float(li) - float(li0)) < 1e-3 and abs(float(l) - float(lw0 + li0)) < 3e-3
    assert gt is data["images_block"] and pred.shape == (32, 4, 4, 3) and depth.shape == (32, 4, 4)
    # whole view (reduced): staged in chunks, ground truth handed back, no decoder on this branch
    H, W = 20, 24
    pose, _, _ = cf.orbit_rays(1, seed=3)
    intr = np.array([30.0, 30.0, W / 2, H / 2], np.float32)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, H, W)
    images = torch.rand(1, H, W, 3)
    whole = {"H": H, "W": W, "rays_o": o.cuda(), "rays_d": d.cuda(), "images": images.cuda()}
    with torch.no_grad():
        pred, depth, gt, decoded, li, lw, l = trainer.eval_step(m, whole, msg.cuda(), dict(kw, max_ray_batch=100), render_whole=True)
        ref = fr.render(o, d, msg, P, S, staged=True, max_ray_batch=100, bg_color=1, dt_gamma=0.0, max_steps=1024)
        t_rgb, t_depth = trainer.test_step(m, whole, msg.cuda(), dict(kw, max_ray_batch=100), bg_color=1)
    assert decoded is None and float(l) == 0.0 and pred.shape == (1, H, W, 3) and depth.shape == (1, H, W) and gt is whole["images"]
    np.testing.assert_allclose(pred.cpu().numpy().reshape(-1, 3), ref["image"].clamp(0, 1).numpy().reshape(-1, 3), rtol=0, atol=1e-3)
    assert torch.equal(t_rgb, pred) and torch.equal(torch.nan_to_num(t_depth), torch.nan_to_num(depth))


def test_training_trajectory_psnr_and_bit_accuracy_track_the_oracle():
    """north_star: "rendered PSNR and 32-bit watermark bit-accuracy matching the reference within 0.1 dB / 1 bit".  Both sides
    train the codebook + decoder for six steps of the reference's loop body (fresh message per step, Adam(0.9, 0.99, eps 1e-15),
    utils_wtmk_disen.py:1164-1181) from the same state -- this repo's kernels on the GPU, the oracle's autograd on the CPU -- and
    are then evaluated the way test_bitacc / test_image do (eval-mode renders): PSNR of the watermarked content render against
    the clean render, bit accuracy of the decoded block renders, for three held-out messages."""
    import copy
    from nerf_signature_amd import trainer
    from nerf_signature_amd.trainer import BIT_ACC, PSNRMeter
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    bo, bd, co, cd, _ = _data(n_content=256, block=4)
    kw = dict(dt_gamma=0.0, max_steps=1024)
    with torch.no_grad():
        clean = fr.render(co, cd, None, P, S, bg_color=1, **kw)["image"].detach()
    # ground truth = the clean render (provider_wtmk.py:415) plus a fixed +-0.03 pattern standing in for the clean model's fit
    # error against real photographs (PSNR ~35 dB, the regime the 0.1 dB criterion is meant for; against the bare clean render
    # the watermark's own 1e-4 perturbation would be compared with this path's 5e-5 rendering tolerance)
    gt = (clean + torch.from_numpy(np.random.RandomState(7).uniform(-0.03, 0.03, clean.shape).astype(np.float32))).clamp(0, 1)
    dec_cpu = copy.deepcopy(m.msg_decoder).cpu()
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    msgs = [torch.from_numpy(np.random.RandomState(40 + s).randint(0, 2, 32).astype(np.float32)) for s in range(6)]
    opt1 = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = trainer.WatermarkLoop(m, opt1, kw)
    opt0 = torch.optim.Adam([{"params": P["cb_tables"]}, {"params": list(dec_cpu.parameters())}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    l1, l0 = [], []
    for msg in msgs:
        out = loop.step(data, msg)
        l1.append([float(out[3].detach()), float(out[4].detach())])
        opt0.zero_grad(set_to_none=True)
        ref = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec_cpu, **kw)
        ref["loss"].backward()
        opt0.step()
        l0.append([float(ref["lossi"].detach()), float(ref["lossw"].detach())])
    l1, l0 = np.array(l1), np.array(l0)
    print("\nloss trajectories (image, watermark): gpu\n", l1, "\noracle\n", l0)
    np.testing.assert_allclose(l1[:, 1], l0[:, 1], rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(l1[:, 0], l0[:, 0], rtol=1e-2, atol=1e-7)
    assert l0[-1, 0] < l0[0, 0] and l1[-1, 0] < l1[0, 0]                                  # training moves the image loss on both sides

    m.eval()
    dec_cpu.eval()
    psnr1, psnr0, acc1, acc0 = PSNRMeter(), PSNRMeter(), BIT_ACC(), BIT_ACC()
    held1, held0 = [], []        # the watermarked content renders of the three held-out messages (used again below)
    with torch.no_grad():
        for s in range(3):
            msg = torch.from_numpy(np.random.RandomState(90 + s).randint(0, 2, 32).astype(np.float32))
            img1 = m.render(co.cuda(), cd.cuda(), msg, staged=False, bg_color=1, perturb=False, **kw)["image"].cpu()
            img0 = fr.render(co, cd, msg, P, S, training=False, bg_color=1, **kw)["image"]
            held1.append(img1.double())
            held0.append(img0.double())
            psnr1.update(img1, gt)
            psnr0.update(img0, gt)
            blk1 = m.render(bo.cuda(), bd.cuda(), msg, staged=False, bg_color=1, perturb=False, **kw)["image"]
            blk0 = fr.render(bo, bd, msg, P, S, training=False, bg_color=1, **kw)["image"]
            d1 = m.msg_decoder(m.normalization(blk1.clamp(0, 1).permute(0, 3, 1, 2))).cpu()
            d0 = dec_cpu(fr.normalize_img(blk0.clamp(0, 1).permute(0, 3, 1, 2)))
            acc1.update(d1.permute(1, 0), msg[None])
            acc0.update(d0.permute(1, 0), msg[None])
    print(f"PSNR gpu {psnr1.measure():.3f} dB / oracle {psnr0.measure():.3f} dB; bit accuracy gpu {acc1.measure():.4f} / oracle {acc0.measure():.4f}")
    assert abs(psnr1.measure() - psnr0.measure()) < 0.1                                    # dB
    # The same criterion where it can fail: the noise pattern in `gt` fixes both PSNRs above near 35 dB whatever the watermark does.
    # PSNR of the watermarked render against each side's OWN clean render measures the watermark's perturbation itself (fp64 MSE):
    # a rendering error of the order of the perturbation moves it by whole dB.
    with torch.no_grad():
        c1 = m.render(co.cuda(), cd.cuda(), None, staged=False, bg_color=1, perturb=False, **kw)["image"].cpu().double()     # (no message: the same for all three)
        c0 = fr.render(co, cd, None, P, S, training=False, bg_color=1, **kw)["image"].double()
        wm1 = [-10 * np.log10(float(((i1 - c1) ** 2).mean())) for i1 in held1]
        wm0 = [-10 * np.log10(float(((i0 - c0) ** 2).mean())) for i0 in held0]
    print("PSNR of the watermarked render vs the clean render (dB): gpu", np.round(wm1, 3), "oracle", np.round(wm0, 3))
    assert all(20.0 < v < 80.0 for v in wm0)                      # a real, visible-in-fp32 perturbation
    np.testing.assert_allclose(wm1, wm0, rtol=0, atol=0.1)        # dB
    assert abs(acc1.measure() - acc0.measure()) <= 1.0 / 32 + 1e-9                         # one bit


def test_uniform_sample_path_run_matches_oracle():
    """BASELINE config 1 shape: NeRFRenderer.run (no occupancy grid): 512 uniform samples per ray, colour where weight > 1e-4."""
    from nerf_signature_amd.network import NeRFNetwork
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    m.cuda_ray = False
    pose, intr, inds = cf.orbit_rays(96, seed=6)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    msg = torch.from_numpy(cf.messages(32)[2])
    with torch.no_grad():
        ref = fr.render(o, d, msg, P, S, cuda_ray=False, num_steps=512, bg_color=1)
        out = m.render(o.cuda(), d.cuda(), msg, staged=False, num_steps=512, upsample_steps=0, bg_color=1, perturb=False, junk=3)
    np.testing.assert_allclose(out["image"].cpu().numpy(), ref["image"].numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), ref["weights_sum"].numpy(), rtol=0, atol=1e-3)
    hit = ~torch.isnan(ref["depth"])
    np.testing.assert_allclose(out["depth"].cpu()[hit].numpy(), ref["depth"][hit].numpy(), rtol=0, atol=1e-3)


def test_step_mirrors_colour_space_and_alpha_handling():
    """train_step / eval_step with `color_space="linear"` (utils_wtmk_disen.py:603-604, 691-692: converted in place) and eval_step's RGBA blend
    against a white background (:695-699); train_step with RGBA images raises like the reference's (whose bg_color is unassigned there, :585-590)."""
    from nerf_signature_amd import trainer
    m, bitfield, C = _model()
    bo, bd, co, cd, gt = _data(n_content=64, block=4)
    msg = torch.from_numpy(cf.messages(32)[1]).cuda()
    kw = dict(dt_gamma=0, max_steps=1024)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda().clone()}}
    want_gt = torch.where(gt < 0.04045, gt / 12.92, ((gt + 0.055) / 1.055) ** 2.4).cuda()
    out = trainer.train_step(m, data, msg, kw, color_space="linear")
    assert torch.allclose(out[1], want_gt, rtol=0, atol=2e-6) and out[1] is data["content"]["images"]          # in place, as the reference does it
    assert abs(float(out[3]) - float(((out[2] - want_gt) ** 2).mean())) < 1e-6
    data["content"]["images"] = torch.rand(1, 64, 4, device="cuda")
    with pytest.raises(NotImplementedError, match="UnboundLocalError"):
        trainer.train_step(m, data, msg, kw)
    H = W = 8
    pose, intr, _ = cf.orbit_rays(1, seed=3)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr * np.array([0.02, 0.02, 0.02, 0.02], np.float32), H, W)
    rgba = torch.rand(1, H, W, 4, device="cuda")
    ev = {"H": H, "W": W, "rays_o": o.cuda(), "rays_d": d.cuda(), "images": rgba.clone()}
    with torch.no_grad():
        pred, depth, gt_rgb, decoded, li, lw, l = trainer.eval_step(m, ev, msg, kw, render_whole=True, color_space="linear")
    lin = torch.where(rgba[..., :3] < 0.04045, rgba[..., :3] / 12.92, ((rgba[..., :3] + 0.055) / 1.055) ** 2.4)
    assert torch.allclose(gt_rgb, lin * rgba[..., 3:] + (1 - rgba[..., 3:]), rtol=0, atol=1e-6) and pred.shape == (1, H, W, 3) and decoded is None


def test_uniform_sample_path_with_importance_resampling_matches_oracle(strict_mlp):
    """NeRFRenderer.run with upsample_steps > 0 (renderer_wtmk.py:166-201, sample_pdf :12-47): 96 coarse + 64 re-sampled depths per ray, merged
    in depth order; the re-sampled points see the CLEAN field (the reference calls density(new_xyzs) without the message, :187).  Eval mode
    (evenly spaced quantiles: deterministic) without gradients against the oracle; then with gradients: image and codebook gradient of an MSE
    loss against the oracle's autograd (gradient flows through the coarse samples only); training mode (random quantiles): finite, and the
    same coarse-only picture within the re-sampling noise."""
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    m.cuda_ray = False
    pose, intr, inds = cf.orbit_rays(64, seed=9)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(np.minimum(inds, 80200 + 5 * np.arange(64)))[None])
    msg = torch.from_numpy(cf.messages(32)[2])
    kw = dict(staged=False, num_steps=96, upsample_steps=64, bg_color=1, perturb=False)
    m.eval()
    with torch.no_grad():
        ref = fr.render(o, d, msg, P, S, cuda_ray=False, training=False, num_steps=96, upsample_steps=64, bg_color=1)
        out = m.render(o.cuda(), d.cuda(), msg, **kw)
        coarse = m.render(o.cuda(), d.cuda(), msg, **dict(kw, upsample_steps=0))
    assert set(out) == {"image", "depth", "weights_sum"}
    np.testing.assert_allclose(out["image"].cpu().numpy(), ref["image"].numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), ref["weights_sum"].numpy(), rtol=0, atol=1e-3)
    hit = ~torch.isnan(ref["depth"])
    np.testing.assert_allclose(out["depth"].cpu()[hit].numpy(), ref["depth"][hit].numpy(), rtol=0, atol=1e-3)
    assert float((out["image"] - coarse["image"]).abs().max()) > 1e-4          # the extra samples do change the picture
    # with gradients
    gt = torch.rand(1, 64, 3)
    ref = fr.render(o, d, msg, P, S, cuda_ray=False, training=False, num_steps=96, upsample_steps=64, bg_color=1)
    ((ref["image"] - gt) ** 2).mean().backward()
    out = m.render(o.cuda(), d.cuda(), msg, **kw)
    ((out["image"] - gt.cuda()) ** 2).mean().backward()
    np.testing.assert_allclose(out["image"].detach().cpu().numpy(), ref["image"].detach().numpy(), rtol=0, atol=1e-3)
    bits = [int(v) for v in msg]
    for i in (0, 7, 31):
        g1, g0 = m.msg_encoder.embeddings[2 * i + bits[i]].weight.grad.cpu(), P["cb_tables"][2 * i + bits[i]].grad
        assert float((g1 - g0).norm() / g0.norm()) < 5e-3
        assert m.msg_encoder.embeddings[2 * i + 1 - bits[i]].weight.grad is None
    # training mode: random quantiles
    m.train()
    with torch.no_grad():
        rnd = m.render(o.cuda(), d.cuda(), msg, **kw)
    assert torch.isfinite(rnd["image"]).all() and float((rnd["image"] - out["image"].detach()).abs().max()) < 0.05


def test_uniform_sample_path_trains_the_codebook(strict_mlp):
    """NeRFRenderer.run with gradients (what `main_nerf_wtmk.py` without --cuda_ray trains through: density() + masked color(),
    renderer_wtmk.py:187-229): image and the codebook gradient of an MSE loss against the oracle's autograd through the same path."""
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    m.cuda_ray = False
    pose, intr, inds = cf.orbit_rays(48, seed=8)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(np.minimum(inds, 80200 + 7 * np.arange(48)))[None])
    msg = torch.from_numpy(cf.messages(32)[2])
    gt = torch.rand(1, 48, 3)
    ref = fr.render(o, d, msg, P, S, cuda_ray=False, num_steps=128, bg_color=1)
    ((ref["image"] - gt) ** 2).mean().backward()
    out = m.render(o.cuda(), d.cuda(), msg, staged=False, num_steps=128, upsample_steps=0, bg_color=1, perturb=False)
    ((out["image"] - gt.cuda()) ** 2).mean().backward()
    np.testing.assert_allclose(out["image"].detach().cpu().numpy(), ref["image"].detach().numpy(), rtol=0, atol=1e-3)
    bits = [int(v) for v in msg]
    g1, g0 = m.msg_encoder.embeddings[bits[0]].weight.grad.cpu(), P["cb_tables"][bits[0]].grad
    assert float(g0.norm()) > 0 and float((g1 - g0).norm() / g0.norm()) < 5e-3
    assert m.msg_encoder.embeddings[1 - bits[0]].weight.grad is None and all(p.grad is None for p in m.encoder.parameters())


def test_density_grid_maintenance():
    """update_extra_state / mark_untrained_grid (renderer_wtmk.py:380-538) on the GPU: bitfield == packbits(grid), EMA rule, -1 marking."""
    from oracle import raymarch_ref as orm
    m, bitfield, C = _model()
    m.density_grid.zero_()
    m.density_bitfield.zero_()
    torch.manual_seed(0)
    m.update_extra_state(message=None)                   # first call: every cell probed
    grid = m.density_grid.cpu().numpy()
    assert m.iter_density == 1 and (grid >= 0).all() and grid.max() > 0
    np.testing.assert_allclose(m.mean_density, float(grid.clip(min=0).mean()), rtol=1e-4)
    thresh = min(m.mean_density, m.density_thresh)
    np.testing.assert_array_equal(m.density_bitfield.cpu().numpy(), orm.packbits(grid, thresh))
    before = m.density_grid.clone()
    m.update_extra_state(message=None)
    after = m.density_grid
    assert bool((after >= before * 0.95 - 1e-6).all())   # max(grid * decay, new sigma)
    poses = torch.from_numpy(cf.orbit_rays(1, seed=0)[0])[None]
    m.mark_untrained_grid(poses, (555.56, 555.56, 200.0, 200.0))
    marked = (m.density_grid == -1)
    assert bool(marked.any()) and not bool(marked.all())  # cells outside the single camera's frustum are marked untrained


def test_clean_render_prepass_matches_oracle():
    """N2: the clean-render pre-pass (message=None, staged, rays generated on the device) vs the oracle's staged render."""
    from nerf_signature_amd import blocks
    m, bitfield, C = _model()
    P, S = _oracle_params(m, bitfield, C)
    H, W = 24, 36
    poses = torch.stack([torch.from_numpy(cf.orbit_rays(1, seed=s)[0]) for s in (0, 1)])
    poses[1, :3, 3] *= 0.9
    intr = np.array([40.0, 40.0, W / 2, H / 2], np.float32)
    imgs = blocks.clean_render(m, poses.cuda(), intr, H, W, dict(dt_gamma=0, max_steps=1024), max_ray_batch=300)
    assert imgs.shape == (2, H, W, 3)
    for b in range(2):
        o, d = fr.get_rays(poses[b:b + 1], intr, H, W)
        ref = fr.render(o, d, None, P, S, staged=True, max_ray_batch=300, bg_color=1, dt_gamma=0.0, max_steps=1024)
        np.testing.assert_allclose(imgs[b].reshape(-1, 3).cpu().numpy(), ref["image"][0].numpy(), rtol=0, atol=1e-3)
    coords, bh, bw = blocks.process_image(imgs[:1].cpu(), 4, 6, 5)
    assert coords.shape == (5, 4) and (bh, bw) == (6, 6)


@pytest.mark.parametrize("shape", [(32, 64, 12, 12), (48, 64, 11, 15), (32, 1, 12, 12), (3, 5, 2, 3), (40, 8, 16, 16)])
def test_fused_batchnorm_gelu_matches_torch(shape):
    """dec_bn_gelu_fwd/_bwd == GELU(BatchNorm2d(eps=1e-3, batch statistics)) of hidden_models.py:24-28, values and all gradients,
    and the block as a whole (conv bias dropped in front of the BatchNorm) == the stock module."""
    from nerf_signature_amd.hidden_models import ConvBNRelu, _BNGelu
    torch.manual_seed(3)
    N, C, H, W = shape
    x = (torch.randn(shape, device="cuda") * 1.7 + 0.4).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gamma = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device="cuda") * 0.3).requires_grad_(True)
    dy = torch.randn(shape, device="cuda")
    y1 = _BNGelu.apply(x, gamma, beta, 1e-3)
    g1 = torch.autograd.grad(y1, (x, gamma, beta), dy)
    xd, gd, bd = (t.detach().double().requires_grad_(True) for t in (x, gamma, beta))
    y0 = torch.nn.functional.gelu(torch.nn.functional.batch_norm(xd, None, None, gd, bd, True, 0.0, 1e-3))
    g0 = torch.autograd.grad(y0, (xd, gd, bd), dy.double())
    np.testing.assert_allclose(y1.detach().cpu().numpy(), y0.detach().cpu().numpy(), rtol=0, atol=2e-5)   # fp32 vs an fp64 reference
    for a, b in zip(g1, g0):
        assert float((a.double() - b).norm() / b.norm()) < 2e-5
    if C > 1:
        blk = ConvBNRelu(3, C).cuda()
        img = torch.randn(N, 3, H, W, device="cuda", requires_grad=True)
        out1 = blk(img)
        out0 = blk.layers(img)     # the stock operator chain, bias included
        np.testing.assert_allclose(out1.detach().cpu().numpy(), out0.detach().cpu().numpy(), rtol=0, atol=2e-5)
        p = [blk.layers[0].weight, blk.layers[1].weight, blk.layers[1].bias, img]
        ga, gb = torch.autograd.grad(out1, p, dy), torch.autograd.grad(out0, p, dy)
        for a, b in zip(ga, gb):
            assert float((a - b).norm() / b.norm()) < 1e-4


@pytest.mark.parametrize("shape", [(32, 3, 12, 12), (48, 3, 11, 15), (5, 3, 7, 5), (2, 6, 16, 16), (4, 3, 3, 37), (3, 3, 1, 40), (16, 3, 5, 5), (64, 3, 12, 16), (1, 3, 8, 8)])
def test_fused_decoder_matches_stock_operators(shape, monkeypatch):
    """dec_forward/dec_backward == the stock PyTorch chain of hidden_models.py:104-137 (Conv2d, BatchNorm2d with batch
    statistics, GELU, AdaptiveAvgPool2d, Linear) evaluated in fp64: decoded bits, the gradient of the image and of every parameter.
    One launch per layer each way, split-bf16 (fp32-class) products -- logits 1e-3, gradients 2e-3 relative L2.  (Rounds 2-3 also carried a
    persistent one-launch-per-direction route with fp16 operands; it never beat this chain -- profiles/r04_rank8_critical_path.txt -- and was
    removed in round 4.)"""
    from nerf_signature_amd.hidden_models import get_hidden_decoder_multi_views
    torch.manual_seed(11)
    B, Cin, H, W = shape
    dec = get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=Cin, channels=64).cuda()
    with torch.no_grad():
        for p in dec.parameters():               # away from the default init: BN weights != 1, biases != 0
            p.add_(0.1 * torch.randn_like(p))
    img = torch.randn(B, Cin, H, W, device="cuda", requires_grad=True)
    gout = torch.randn(B, 1, device="cuda")
    assert dec._fused_params(*img.shape, img) is not None
    out1 = dec(img)
    g1 = torch.autograd.grad(out1, [img] + [p for p in dec.parameters()], gout, allow_unused=True)
    torch.cuda.synchronize()
    monkeypatch.setenv("NERFSIG_DECODER", "torch")
    dec64 = dec.double()
    img64 = img.detach().double().requires_grad_(True)
    out0 = dec64.layers(img64).squeeze(-1).squeeze(-1)
    out0 = dec64.linear(out0)
    g0 = torch.autograd.grad(out0, [img64] + [p for p in dec64.parameters()], gout.double())
    atol, rtol_g = 1e-3, 2e-3
    err_out = float((out1.detach().double() - out0).abs().max())
    assert err_out < atol, err_out
    names = ["img"] + [n for n, _ in dec64.named_parameters()]
    for n, a, b in zip(names, g1, g0):
        if a is None:                             # conv bias in front of a BatchNorm: identically zero
            assert n.endswith("layers.0.bias") and float(b.abs().max()) < 1e-9 * max(1.0, float(gout.abs().max()))
            continue
        rel = float((a.double() - b).norm() / (b.norm() + 1e-30))
        assert rel < rtol_g, (n, rel)


def test_fused_finish_and_loss_match_stock_operators():
    """rm_finish_* == renderer_wtmk.py:316-319 and wm_loss_* == utils_wtmk_disen.py:615-640 (MSE + BCE-with-logits(10x) + weighted
    sum), values and gradients, against the stock operator chains in fp64."""
    from nerf_signature_amd.renderer import _Finish
    from nerf_signature_amd.trainer import _WatermarkLoss
    torch.manual_seed(5)
    N = 1000
    image, depth, ws = (torch.rand(N, 3, device="cuda").requires_grad_(True), (torch.rand(N, device="cuda") * 4).requires_grad_(True),
                        torch.rand(N, device="cuda").requires_grad_(True))
    nears, fars = torch.rand(N, device="cuda") + 0.5, torch.rand(N, device="cuda") + 3.0
    for bg in (torch.full((3,), 1.0, device="cuda"), torch.rand(N, 3, device="cuda")):
        gi, gd = torch.randn(N, 3, device="cuda"), torch.randn(N, device="cuda")
        o1 = _Finish.apply(image, depth, ws, nears, fars, bg)
        g1 = torch.autograd.grad(o1, (image, depth, ws), (gi, gd))
        i64, d64, w64 = (t.detach().double().requires_grad_(True) for t in (image, depth, ws))
        o0 = (i64 + (1 - w64).unsqueeze(-1) * bg.double(), torch.clamp(d64 - nears.double(), min=0) / (fars - nears).double())
        g0 = torch.autograd.grad(o0, (i64, d64, w64), (gi.double(), gd.double()))
        for a, b in zip(o1 + g1, o0 + g0):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    content, gt = torch.rand(1, 4096, 3, device="cuda").requires_grad_(True), torch.rand(1, 4096, 3, device="cuda")
    decoded = (torch.randn(32, 1, device="cuda") * 3).requires_grad_(True)     # 10x -> logits up to +-90: the stable form matters
    keys = (torch.rand(32, 1, device="cuda") > 0.5).float()
    for sel in range(3):
        out1 = _WatermarkLoss.apply(content, gt, decoded, keys, 0.7, 1.3, 10.0)
        g1 = torch.autograd.grad(out1[sel], (content, decoded))
        c64, d64 = content.detach().double().requires_grad_(True), decoded.detach().double().requires_grad_(True)
        li = ((c64 - gt.double()) ** 2).mean()
        lw = torch.nn.functional.binary_cross_entropy_with_logits(d64 * 10.0, keys.double(), reduction="mean")
        out0 = (li, lw, 0.7 * lw + 1.3 * li)
        g0 = torch.autograd.grad(out0[sel], (c64, d64), allow_unused=True)
        np.testing.assert_allclose([float(v) for v in out1], [float(v) for v in out0], rtol=2e-6)
        for a, b in zip(g1, g0):
            b = torch.zeros_like(a, dtype=torch.float64) if b is None else b
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-9)


def test_fused_decoder_on_rendered_blocks_matches_clamp_permute_normalize():
    """decode_rendered == msg_decoder(normalize_img(clamp(image, 0, 1).permute(0, 3, 1, 2))) of utils_wtmk_disen.py:599-603,
    including the gradient through the clamp (zero outside [0, 1])."""
    from nerf_signature_amd.hidden_models import get_hidden_decoder_multi_views, normalize_img
    torch.manual_seed(2)
    dec = get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=3, channels=64).cuda()
    image = (torch.rand(32, 12, 12, 3, device="cuda") * 1.4 - 0.2).requires_grad_(True)   # some values outside [0, 1]
    gout = torch.randn(32, 1, device="cuda")
    out1, pred1 = dec.decode_rendered(image)
    g1 = torch.autograd.grad(out1, [image] + [p for n, p in dec.named_parameters() if not n.endswith("layers.0.bias")], gout)
    dec64, im64 = dec.double(), image.detach().double().requires_grad_(True)
    pred0 = torch.clamp(im64, min=0, max=1)
    x = (pred0.permute(0, 3, 1, 2) - torch.tensor([0.485, 0.456, 0.406], dtype=torch.float64, device="cuda").view(-1, 1, 1)) / \
        torch.tensor([0.229, 0.224, 0.225], dtype=torch.float64, device="cuda").view(-1, 1, 1)
    out0 = dec64.linear(dec64.layers(x).squeeze(-1).squeeze(-1))
    g0 = torch.autograd.grad(out0, [im64] + [p for n, p in dec64.named_parameters() if not n.endswith("layers.0.bias")], gout.double())
    assert torch.equal(pred1.double(), pred0.detach()) and not pred1.requires_grad
    np.testing.assert_allclose(out1.detach().cpu().numpy(), out0.detach().cpu().numpy(), rtol=0, atol=1e-3)
    outside = (image.detach() < 0) | (image.detach() > 1)
    assert outside.any() and float(g1[0][outside].abs().max()) == 0.0
    for a, b in zip(g1, g0):
        assert float((a.double() - b).norm() / b.norm()) < 2e-3


def test_dense_adam_matches_torch_adam():
    """opt_adam_dense == torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15) on tensors of odd sizes over several steps, a tensor
    skipped for one step (grad None) included."""
    from nerf_signature_amd.optim import CodebookAdam
    torch.manual_seed(1)
    shapes = [(64, 3, 3, 3), (64,), (64, 64, 3, 3), (1, 64, 3, 3), (1,), (1, 1), (1030,)] * 6   # 42 tensors: two launches of <= 32
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = CodebookAdam(a, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, capturable=True)
    ob = torch.optim.Adam(b, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    lr_dev = torch.tensor(1e-2, device="cuda")
    for step in range(5):
        for i, (p, q) in enumerate(zip(a, b)):
            g = torch.randn_like(p) * (10.0 ** (i % 5 - 4))
            skip = step == 2 and i == 3
            p.grad, q.grad = (None, None) if skip else (g.clone(), g.clone())
        oa.step_dense(lr_dev)
        ob.step()
    for i, (p, q) in enumerate(zip(a, b)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=str(i))
        assert float(oa.state[p]["step"]) == float(ob.state[q]["step"])
        np.testing.assert_allclose(oa.state[p]["exp_avg_sq"].cpu().numpy(), ob.state[q]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=0)


def test_dense_adam_with_a_large_tensor_in_the_group_matches_torch_adam():
    """A group that holds a large tensor runs the wide launch, and its medium-sized tensors (>= 1024 elements, a multiple of 4: stage 1's two MLP parameter
    vectors beside the 4 MiB tables) ride along in it; the small and the odd-sized ones keep the element-wise launch.  All against torch.optim.Adam."""
    from nerf_signature_amd.optim import CodebookAdam
    torch.manual_seed(2)
    shapes = [(1 << 17,), (3072,), (7168,), (1030,), (64,), (1 << 16, 2), (1024,), (5, 4097)]
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa = CodebookAdam(a, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, capturable=True)
    ob = torch.optim.Adam(b, lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    lr_dev = torch.tensor(1e-2, device="cuda")
    for step in range(4):
        for i, (p, q) in enumerate(zip(a, b)):
            g = torch.randn_like(p) * (10.0 ** (i % 5 - 4))
            p.grad, q.grad = g.clone(), g.clone()
        oa.step_dense(lr_dev)
        ob.step()
    for i, (p, q) in enumerate(zip(a, b)):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=str(i))
        ma = ob.state[q]["exp_avg"].cpu().numpy()      # (torch forms it as a lerp: elements near zero differ by rounding of the other association)
        np.testing.assert_allclose(oa.state[p]["exp_avg"].cpu().numpy(), ma, rtol=2e-6, atol=1e-6 * float(np.abs(ma).max()), err_msg=str(i))
        np.testing.assert_allclose(oa.state[p]["exp_avg_sq"].cpu().numpy(), ob.state[q]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=0, err_msg=str(i))


def test_fused_decoder_gradients_are_views_of_one_flat_buffer():
    """What dp.GradExchange relies on for its single in-place all-reduce: after backward() every decoder parameter's .grad is a
    view of one contiguous buffer that they tile exactly (conv biases, whose gradient is identically zero, have none)."""
    from nerf_signature_amd.dp import _common_base
    from nerf_signature_amd.hidden_models import get_hidden_decoder_multi_views
    torch.manual_seed(4)
    dec = get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=3, channels=64).cuda()
    image = torch.rand(32, 12, 12, 3, device="cuda", requires_grad=True)
    out, _ = dec.decode_rendered(image)
    out.sum().backward()
    grads = [p.grad for p in dec.parameters() if p.grad is not None]
    assert len(grads) == 29 and sum(p.grad is None for p in dec.parameters()) == 9
    flat = _common_base(grads)
    assert flat is not None and flat.numel() == sum(g.numel() for g in grads)
    before = [g.clone() for g in grads]
    flat.mul_(2.0)                                   # what the exchange does in place reaches every .grad
    assert all(torch.equal(g, 2.0 * b) for g, b in zip(grads, before))


def test_fused_staged_render_equals_chunked(monkeypatch):
    """render(staged=True) under no_grad on the training path: many max_ray_batch chunks per launch sequence (one count read-back)
    give bit-identical images and depths, and leave `local_step` and the `step_counter` ring as the reference's per-chunk calls
    (renderer_wtmk.py:555-570, 282-284) leave them."""
    m, _, _ = _model()
    pose, intr, inds = cf.orbit_rays(10000, seed=4)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    o, d = o.cuda(), d.cuda()
    msg = torch.from_numpy(cf.messages(32)[2])
    kw = dict(staged=True, max_ray_batch=1024, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)
    monkeypatch.setattr(type(m), "STAGED_SUPER_RAYS", 4096)          # three launch sequences: 4 + 4 + 2 chunks (the last one ragged)
    results = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("NERFSIG_STAGED_FUSED", fused)
        m.local_step = 5
        m.step_counter.zero_()
        with torch.no_grad():
            out = m.render(o, d, msg, **kw)
        results[fused] = (out["image"].clone(), out["depth"].clone(), m.local_step, m.step_counter.clone())
    (i0, d0, s0, c0), (i1, d1, s1, c1) = results["0"], results["1"]
    assert torch.equal(i0, i1) and torch.equal(torch.nan_to_num(d0), torch.nan_to_num(d1))
    assert s0 == s1 == 5 + 10 and torch.equal(c0, c1) and int(c0[:, 1].sum()) > 0
    with torch.enable_grad():                                        # with gradients enabled the chunked route is taken (and still agrees)
        monkeypatch.setenv("NERFSIG_STAGED_FUSED", "1")
        out = m.render(o, d, msg, **kw)
    assert torch.equal(out["image"].detach(), i0)


@pytest.mark.parametrize("D", [16, 48])
def test_other_message_lengths_train_step_and_captured_loop(D):
    """--message_dim is a CLI option of the reference (main_nerf_wtmk.py); the north_star names 32 and (config 5) 48 bits.  One train_step at
    D bits against the oracle (point counts exact, images / losses 1e-3, shared gradient in aggregate), then the captured loop against the eager
    loop for three steps at that D (96 table pointers at D = 48 go through the by-value pointer tables of the pre-sum / Adam kernels)."""
    import copy
    from nerf_signature_amd import trainer
    from nerf_signature_amd.optim import CodebookAdam
    m, bitfield, C = _model(D=D)
    P, S = _oracle_params(m, bitfield, C)
    bo, bd, co, cd, gt = _data(n_content=200, block=5)
    reps = (D + 31) // 32
    bo, bd = torch.cat([bo] * reps)[:D].contiguous(), torch.cat([bd + 0.002 * k for k in range(reps)])[:D].contiguous()
    bd = torch.nn.functional.normalize(bd, dim=-1)
    msg = torch.from_numpy(cf.messages(D)[2])
    dec_cpu = copy.deepcopy(m.msg_decoder).cpu()
    ref = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec_cpu, dt_gamma=0.0, max_steps=1024)
    ref["loss"].backward()
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    out = trainer.train_step(m, data, msg, dict(dt_gamma=0, max_steps=1024))
    out[5].backward()
    assert int(m.step_counter[0, 0]) == ref["block"]["n_points"] and int(m.step_counter[1, 0]) == ref["content"]["n_points"]
    assert out[0].shape == (D, 5, 5, 3)
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), ref["pred_rgb"].detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(out[2].detach().cpu().numpy(), ref["content_pred_rgb"].detach().numpy(), rtol=0, atol=1e-3)
    for k, name in ((3, "lossi"), (4, "lossw"), (5, "loss")):
        np.testing.assert_allclose(float(out[k].detach()), float(ref[name].detach()), rtol=1e-3, atol=1e-3)
    bits = [int(v) for v in msg]
    for i in (0, D - 1):      # first and last bit: selected table has the shared gradient, its partner none
        g1, g0 = m.msg_encoder.embeddings[2 * i + bits[i]].weight.grad, P["cb_tables"][2 * i + bits[i]].grad
        assert float((g1.cpu() - g0).norm() / g0.norm()) < 5e-2
        assert m.msg_encoder.embeddings[2 * i + 1 - bits[i]].weight.grad is None
    # captured loop == eager loop at this D
    msgs = [torch.from_numpy(np.random.RandomState(40 + s).randint(0, 2, D).astype(np.float32)) for s in range(3)]
    losses = {}
    for graphed in (False, True):
        torch.manual_seed(0)
        m2, _, _ = _model(D=D)
        opt = CodebookAdam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, capturable=graphed)
        if graphed:
            loop = trainer.GraphedWatermarkLoop(m2, opt, dict(dt_gamma=0, max_steps=1024), data)
            held = [loop.step(msg_k, next_message=msgs[k + 1] if k + 1 < 3 else None)[5].detach().clone() for k, msg_k in enumerate(msgs)]
            losses[graphed] = [float(v) for v in held]
            assert not loop.overflowed()
        else:
            loop = trainer.WatermarkLoop(m2, opt, dict(dt_gamma=0, max_steps=1024))
            losses[graphed] = [float(loop.step(data, msg_k)[5].detach()) for msg_k in msgs]
    np.testing.assert_allclose(losses[True], losses[False], rtol=2e-3, atol=2e-4)


def test_finetune_decoder_mode_trains_the_decoder_only():
    """NeRFNetwork(finetune_decoder=True) (network_wtmk_tcnn.py:90-95, :180-181): the codebook is frozen too, get_params exposes the decoder
    alone; a training step must still run -- through train_step's fused loss path and through the loops -- and move only the decoder."""
    import copy
    from nerf_signature_amd import trainer
    from nerf_signature_amd.network import NeRFNetwork
    m0, bitfield, C = _model()
    m = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1, finetune_decoder=True)
    m.load_state_dict(m0.state_dict())
    m.cuda().train()
    assert not any(p.requires_grad for p in m.msg_encoder.parameters()) and len(m.get_params(1e-2)) == 1
    P, S = _oracle_params(m, bitfield, C)
    bo, bd, co, cd, gt = _data(n_content=128, block=4)
    msg = torch.from_numpy(cf.messages(32)[2])
    dec_cpu = copy.deepcopy(m.msg_decoder).cpu()
    ref = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec_cpu, dt_gamma=0.0, max_steps=1024)
    ref["lossw"].backward()          # (the image loss has no trainable input in this mode)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt.cuda()}}
    out = trainer.train_step(m, data, msg, dict(dt_gamma=0, max_steps=1024))
    trainer.backward_from_loss_kernel(out)
    np.testing.assert_allclose(float(out[5].detach()), float(ref["loss"].detach()), rtol=1e-3, atol=1e-3)
    assert all(e.weight.grad is None for e in m.msg_encoder.embeddings)
    keep = [n for n, _ in m.msg_decoder.named_parameters() if not n.endswith("layers.0.bias")]
    p1, p0 = dict(m.msg_decoder.named_parameters()), dict(dec_cpu.named_parameters())
    v1 = torch.cat([p1[n].grad.reshape(-1).cpu() for n in keep])
    v0 = torch.cat([p0[n].grad.reshape(-1) for n in keep])
    assert float((v1 - v0).norm() / v0.norm()) < 5e-3
    # the eager loop with the reference's optimiser construction
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = trainer.WatermarkLoop(m, opt, dict(dt_gamma=0, max_steps=1024))
    before = torch.cat([p.detach().reshape(-1).clone() for p in m.msg_decoder.parameters()])
    tables = [e.weight.detach().clone() for e in m.msg_encoder.embeddings[:4]]
    l = [float(loop.step(data, msg)[4].detach()) for _ in range(3)]
    after = torch.cat([p.detach().reshape(-1) for p in m.msg_decoder.parameters()])
    assert not torch.equal(before, after) and all(torch.equal(a, e.weight) for a, e in zip(tables, m.msg_encoder.embeddings[:4])) and l[-1] < l[0]


@pytest.mark.parametrize("bound,dt_gamma,T_thresh,training", [(1.0, 0.0, 1e-2, True), (1.0, 0.0, 1e-2, False), (2.0, 1 / 128, 1e-4, True),
                                                              (2.0, 1 / 128, 1e-3, False), (1.0, 0.0, 1e-4, False)])
def test_opaque_field_early_termination_through_render(bound, dt_gamma, T_thresh, training, strict_mlp):
    """A field dense enough to end rays early (sigma head's log-density row scaled up): the `T < T_thresh` exits of the training compositor
    (raymarching.cu:542-557: the sample that crosses the threshold is still accumulated) and of the eval loop's bursts (:868, alive-list
    compaction) at the level of render(), for several thresholds / cascades / cone angles, against the oracle.  Tolerance 3e-3 instead of
    the path's 1e-3: sigma = exp(40 h0) multiplies the MLP's rounding error in h0 by 40 (this test is about where rays stop, the MLP's
    accuracy has its own tests)."""
    tol = 3e-3
    m, P, S = _oracle_and_model(32, bound)
    with torch.no_grad():
        w = m.sigma_net.params.detach().clone()
        w2 = w[2048:].view(16, 64)
        w2[0] *= 40.0                                            # row 0 = log-density: sigma = exp(40 h0) is opaque wherever h0 > 0
        m.sigma_net.params.copy_(w)
    P["sigma_params"] = m.sigma_net.params.detach().cpu()
    pose, intr, inds = cf.orbit_rays(160, seed=12, radius=3.2248 if bound == 1.0 else 1.3)
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, torch.from_numpy(inds)[None])
    msg = torch.from_numpy(cf.messages(32)[2])
    kw = dict(bg_color=1, dt_gamma=dt_gamma, max_steps=1024, T_thresh=T_thresh)
    with torch.no_grad():
        ref = fr.render(o, d, msg, P, S, training=training, **kw)
        m.train(training)
        out = m.render(o.cuda(), d.cuda(), msg, staged=False, perturb=False, force_all_rays=True, **kw)
    ws = ref.get("weights_sum")
    np.testing.assert_allclose(out["image"].cpu().numpy(), ref["image"].numpy(), rtol=0, atol=tol)
    hit = ~torch.isnan(ref["depth"])
    np.testing.assert_allclose(out["depth"].cpu()[hit].numpy(), ref["depth"][hit].numpy(), rtol=0, atol=tol)
    if training:
        np.testing.assert_allclose(out["weights_sum"].cpu().numpy(), ws.numpy(), rtol=0, atol=tol)
        assert float((ws > 1 - 2 * T_thresh).float().mean()) > 0.02        # some rays really are opaque: they ended on the threshold


@pytest.mark.parametrize("case", ["thin", "opaque", "counter"])
def test_eval_loop_on_device_equals_the_host_driven_loop(case, monkeypatch):
    """renderer_wtmk.py:335-372 with the loop's control on the device (rm_eval_*: no per-round read-back of the survivor count; launches sized for the worst
    case that early-out on device counts; field_fwd_rows over a device row count) against the round-by-round form that reads n_alive back every round:
    the SAME bursts -- (n_alive, n_step) and the alive list of every round -- and bit-identical weights_sum / depth / image, for a thin field (rays live
    until they leave the box), an opaque one (early termination by T_thresh) and the two-cascade scene with dt_gamma > 0."""
    from nerf_signature_amd import raymarching, synthetic
    from nerf_signature_amd.network import NeRFNetwork
    scene = "counter" if case == "counter" else "hotdog"
    cfg = synthetic.SCENES[scene]
    torch.manual_seed(0)
    m = NeRFNetwork(bound=cfg["bound"], cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=cfg["message_dim"], n_views=1)
    synthetic.init_model(m, scene, opaque=(case == "opaque"))
    m.cuda().eval()
    o, d = synthetic.content_rays(scene, 3000, seed=3, device="cuda")
    msg = torch.from_numpy(np.random.RandomState(1).randint(0, 2, cfg["message_dim"]).astype(np.float32))
    kw = dict(dt_gamma=1.0 / 128 if case == "counter" else 0.0, max_steps=1024, bg_color=1, perturb=False, staged=False)
    rounds_host = []
    real = raymarching.march_rays

    def spy(n_alive, n_step, rays_alive, *a, **k):
        rounds_host.append((int(n_alive), int(n_step), rays_alive[:n_alive].clone()))
        return real(n_alive, n_step, rays_alive, *a, **k)

    with torch.no_grad():
        monkeypatch.setenv("NERFSIG_EVAL_LOOP", "host")
        monkeypatch.setattr(raymarching, "march_rays", spy)
        host = m.render(o, d, msg, **kw)
        monkeypatch.setattr(raymarching, "march_rays", real)
        monkeypatch.setenv("NERFSIG_EVAL_LOOP", "device")
        dev = m.render(o, d, msg, **kw)
        # ... and round by round
        prefix, fo_, fd_ = m._flatten_rays(o, d)
        nears, fars = raymarching.near_far_from_aabb(fo_, fd_, m.aabb_infer, m.min_near)
        trace = []
        ws, dp, im = m._eval_loop_on_device(fo_, fd_, msg, nears, fars, kw["dt_gamma"], False, 1024, 1e-4, trace=trace)
    assert len(rounds_host) > (3 if case == "opaque" else 30)
    for k in ("image", "depth"):
        assert torch.equal(torch.nan_to_num(host[k]), torch.nan_to_num(dev[k])), k
    live = [t for t in trace if t[0] > 0]
    assert len(live) == len(rounds_host)
    for (na0, ns0, ids0), (na1, ns1, ids1) in zip(rounds_host, live):
        assert (na0, ns0) == (na1, ns1) and torch.equal(ids0, ids1)
    assert float(ws.max()) <= 1.0 + 1e-5 and bool(torch.isfinite(ws).all())


def test_premarch_hands_the_same_samples_to_the_next_render_once():
    """NeRFRenderer.premarch: the training march of a ray pair issued ahead of its render (trainer.train_step does it for the content rays of an eager step, in
    front of the block render).  The render that follows with the same tensor objects uses those samples -- same image, depth and gradient bit for bit, no
    second march (the counter ring advances once) -- exactly once; other tensors, an in-place change of the rays, other march arguments or a changed
    grid march again as usual."""
    m, _, _ = _model()
    _, _, co, cd, _ = _data(n_content=300)
    o, d = co.cuda(), cd.cuda()
    msg = torch.from_numpy(cf.messages(32)[2]).cuda()
    kw = dict(staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024)

    def render(oo, dd):
        m.zero_grad()
        out = m.render(oo, dd, msg, **kw)
        (out["image"] * torch.linspace(0.5, 1.5, out["image"].numel(), device="cuda").view_as(out["image"])).sum().backward()
        bits = [int(v) for v in msg.cpu()]
        return out["image"].detach().clone(), out["depth"].detach().clone(), m.msg_encoder.embeddings[bits[0]].weight.grad.clone()

    plain = render(o, d)
    step0 = m.local_step
    m.premarch(o, d, 0, 1024)
    assert m._premarched is not None and m.local_step == step0 + 1
    ahead = render(o, d)
    assert m._premarched is None and m.local_step == step0 + 1            # used, and no second march
    for a, b in zip(plain[:2], ahead[:2]):      # (rays that miss the box have depth 0 / 0, as in the reference)
        assert torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0))
    assert float((plain[2] - ahead[2]).abs().max()) <= 1e-6 * float(plain[2].abs().max())      # (float atomics of the small-launch scatter)
    render(o, d)
    assert m.local_step == step0 + 2                                         # once only
    m.premarch(o, d, 0, 1024)
    o2 = o.clone()
    other = render(o2, d)                                                    # other tensor objects: marches itself, the stash is dropped
    assert m._premarched is None and torch.equal(other[0], plain[0])
    m.premarch(o, d, 0, 1024)
    o.add_(0.0)                                                              # an in-place write bumps the version
    before = m.local_step
    render(o, d)
    assert m.local_step == before + 1
    m.premarch(o, d, 0, 1024)
    before = m.local_step
    m.render(o, d, msg, **dict(kw, max_steps=512))
    assert m.local_step == before + 1


def test_standalone_density_and_color_are_differentiable(strict_mlp):
    """VERDICT round 4, missing #4: NeRFNetwork.density / .color called directly under autograd (network_wtmk_tcnn.py:126-176; the reference reaches them that way only from
    `run`).  loss = <sigma, a> + <geo_feat, b> + <color(d, geo_feat, mask), c>: values against the oracle's density / color, the gradient of every SELECTED codebook table
    against the oracle's autograd (one shared gradient), unselected tables untouched; the masked form of color() (renderer_wtmk.py:222-229) zeroes rows and their gradient."""
    m, bitfield, C = _model()
    P, _ = _oracle_params(m, bitfield, C)
    rng = np.random.RandomState(21)
    M = 6007
    x = torch.from_numpy((rng.rand(M, 3) * 1.6 - 0.8).astype(np.float32))
    d = torch.from_numpy(cf.unit_dirs(M, seed=22))
    msg = torch.from_numpy(cf.messages(32)[1])
    a, b, c = (torch.from_numpy(rng.randn(*s).astype(np.float32)) for s in ((M,), (M, 15), (M, 3)))
    mask = torch.from_numpy(rng.rand(M) < 0.7)
    # oracle
    dn = fr.density(x, msg, P)
    col = torch.zeros(M, 3)
    col[mask] = fr.color(d[mask], dn["geo_feat"][mask], P)
    ((dn["sigma"] * a).sum() + (dn["geo_feat"] * b).sum() + (col * c).sum()).backward()
    bits = [int(v) for v in msg]
    want = P["cb_tables"][bits[0]].grad
    # product
    for e in m.msg_encoder.embeddings:
        e.weight.grad = None
    out = m.density(x.cuda(), msg)
    assert out["sigma"].requires_grad and out["geo_feat"].requires_grad
    rgb = m.color(x.cuda(), d.cuda(), mask=mask.cuda(), geo_feat=out["geo_feat"])
    np.testing.assert_allclose(out["sigma"].detach().cpu().numpy(), dn["sigma"].detach().numpy(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(out["geo_feat"].detach().cpu().numpy(), dn["geo_feat"].detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), col.detach().numpy(), rtol=0, atol=1e-3)
    assert float(rgb.detach()[~mask.cuda()].abs().max()) == 0.0
    ((out["sigma"] * a.cuda()).sum() + (out["geo_feat"] * b.cuda()).sum() + (rgb * c.cuda()).sum()).backward()
    for k, e in enumerate(m.msg_encoder.embeddings):
        selected = (k % 2) == bits[k // 2]
        assert (e.weight.grad is not None) == selected, k
    got = m.msg_encoder.embeddings[bits[0]].weight.grad.cpu()
    assert torch.equal(m.msg_encoder.embeddings[2 + bits[1]].weight.grad.cpu(), got)                     # every selected table carries the same gradient
    rel = float((got - want).norm() / want.norm())
    assert rel < 5e-3, rel
    assert torch.equal(got != 0, want != 0) or float(((got != 0) != (want != 0)).float().mean()) < 1e-4    # the same rows are touched
    # without gradients nothing changes: same values, no graph
    with torch.no_grad():
        plain = m.density(x.cuda(), msg)
    assert torch.equal(plain["sigma"], out["sigma"].detach()) and not plain["sigma"].requires_grad
