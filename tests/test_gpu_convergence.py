"""Does the measured path learn the watermark -- and does every way of driving it learn the SAME one?  (north_star: "rendered PSNR and 32-bit
watermark bit-accuracy matching the reference within 0.1 dB / 1 bit"; BASELINE metric "PSNR + bit-acc".)

  (a)+(b) the reference's whole run at bench size (README.md:45: 1000 iterations, lambda_w 0.005, lambda_i 1, lr 1e-2 * 0.1 ** (it / iters)) through
          the captured loop exactly as bench.py drives it, twice (the run-to-run spread of ONE binary: atomics' order), with the block rays declared
          constant, through the eager loop, and through a world-size-1 RCCL group with the sharded optimiser in both execution modes -- then
          Trainer.test_bitacc over 200 random messages and Trainer.test_image against the clean views (nerf_signature_amd/quality.py);
  (c)     a 200-step trajectory tracked step for step by the CPU oracle from a common warm state in which the decoder is already off the
          chance plateau (the plateau is a saddle: WHEN a run leaves it is decided by rounding, so a cold start cannot be tracked), 64 content rays
          + 32 blocks of 4x4 rays, evaluated on held-out messages on both sides."""
import copy
import os

import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr

pytestmark = pytest.mark.gpu


def _expected_adam_steps(D, steps, seed=1234):
    """Per-table step counts after `steps` steps: table 2i + b is stepped whenever bit i of the step's message is b (torch.optim.Adam skips grad=None)."""
    from nerf_signature_amd import quality
    msgs = torch.stack(quality.messages(D, steps + 1, seed)[:steps])
    ones = msgs.sum(0)
    out = []
    for i in range(D):
        out += [float(steps - ones[i]), float(ones[i])]
    return out


def _run_modes(recs, steps, **train_kw):
    """Every execution mode on a fresh stage each: the captured loop twice, fixed blocks, eager, and the world-size-1 RCCL group with the sharded
    optimiser, segmented and with the collectives captured."""
    import torch.distributed as dist
    from nerf_signature_amd import dp, quality

    def one(name, mode):
        stage = quality.watermark_stage("hotdog")
        before = quality.test_bitacc(stage, 40)[0]
        rec = quality.train(stage, steps, mode, **train_kw)
        acc, wrong_mean, wrong_max = quality.test_bitacc(stage, 200)
        psnr = quality.test_image(stage)
        recs[name] = dict(rec, bit_acc=acc, wrong_mean=wrong_mean, wrong_max=wrong_max, psnr=psnr, before=before, checksums=quality.state_checksums(stage),
                          tensors=quality.trained_tensors(stage) if name.startswith("graphed") else None)
        print(f"[{name:>14}] bit acc {before:.3f} -> {acc:.5f} (mean wrong bits/message {wrong_mean:.3f}, worst {wrong_max}); PSNR vs clean views {psnr:.3f} dB; "
              f"{rec['ms_per_step']:.3f} ms/step incl. {len(rec['log'])} host reads; overflow {rec['overflowed']}; loss_i {rec['loss_image']:.3e} loss_w {rec['loss_watermark']:.4f}")
        del stage
        torch.cuda.empty_cache()

    one("graphed", "graphed")
    one("graphed again", "graphed")
    one("fixed blocks", "fixed")
    one("eager", "eager")
    assert not dist.is_initialized()
    os.environ.update(NERFSIG_FORCE_EXCHANGE="1", NERFSIG_SHARD_OPTIMIZER="1")
    try:
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{29500 + os.getpid() % 200}", rank=0, world_size=1)
        assert dp.exchange_active() and dp.optimizer_shard(32) == (0, 32)
        os.environ["NERFSIG_CAPTURE_COLLECTIVES"] = "0"
        one("rccl segmented", "rccl1")
        os.environ["NERFSIG_CAPTURE_COLLECTIVES"] = "1"
        one("rccl captured", "rccl1")
    finally:
        for k in ("NERFSIG_FORCE_EXCHANGE", "NERFSIG_SHARD_OPTIMIZER", "NERFSIG_CAPTURE_COLLECTIVES"):
            os.environ.pop(k, None)
        if dist.is_initialized():
            dist.destroy_process_group()


def test_bench_size_training_converges_the_same_in_every_execution_mode():
    steps, D = 1000, 32
    recs = {}
    _run_modes(recs, steps)
    expected = _expected_adam_steps(D, steps)
    base, again = recs["graphed"], recs["graphed again"]
    # The measured path is bit-reproducible (round 6: the slice owners' replicas merge their fixed-point sums exactly, csrc/hashgrid.hip k_scatter_binned -- the
    # determinism of the reference's embedding_dense_backward, hash_encoding_wtmk_bit.py:99-116): two runs of one binary from one seed leave the SAME codebook and
    # decoder, bit for bit.  (Round 5 asserted a 0.5 dB bound on their PSNR difference instead and measured 0.79 dB on the driver's box: float atomics moved G by
    # half an ulp, and the cold start's saddle amplified it.)
    differing = [a[0] for a, b in zip(base["checksums"], again["checksums"]) if a != b]
    print(f"captured loop twice from one seed: {len(base['checksums'])} trained tensors, {len(differing)} differ; PSNR {base['psnr']:.4f} / {again['psnr']:.4f} dB")
    assert not differing, differing[:8]
    unequal = [na for (na, a), (nb, b) in zip(base["tensors"], again["tensors"]) if na != nb or not torch.equal(a, b)]
    assert len(base["tensors"]) >= 2 * D + 10 and not unequal, unequal[:8]           # torch.equal on every codebook table and every decoder parameter / buffer
    base["tensors"] = again["tensors"] = None
    torch.cuda.empty_cache()
    assert again["psnr"] == base["psnr"] and again["bit_acc"] == base["bit_acc"] and again["log"] == base["log"]
    for name, r in recs.items():
        assert 0.35 < r["before"] < 0.65, (name, r["before"])                # untrained: chance
        assert r["bit_acc"] >= 1.0 - 1.0 / 32, (name, r["bit_acc"])          # trained: on average less than one wrong bit of 32 ...
        assert abs(r["bit_acc"] - base["bit_acc"]) <= 1.0 / 32               # ... and every mode within one bit of the measured path
        assert r["wrong_max"] <= 2, (name, r["wrong_max"])
        assert not r["overflowed"] and r["recaptures"] == 0, name            # no replay dropped a ray
        assert r["adam_steps"] == expected, name                             # per-table Adam step counts: one per selection, in every mode
        assert r["loss_image"] < 5e-6 and r["loss_watermark"] < 0.05, name
        # PSNR of the watermarked views against the clean views = the watermark's own amplitude (MSE ~1e-6).  The OTHER modes run the same arithmetic in another
        # order somewhere (the eager loop's float reductions, the sharded optimiser's pre-sum), and from a COLD start every run first sits on the decoder's chance
        # plateau -- a saddle: WHEN a run leaves it is decided by the last bits of G -- so modes end up to 0.9 dB apart (GPUTEST_r05: 60.51 .. 61.42 dB over six
        # runs; 0.79 dB between two runs of the then non-reproducible captured loop).  2 dB (a quarter in amplitude) is twice that spread: it bounds a SYSTEMATIC
        # difference; the 0.1 dB criterion is applied where it is decidable -- from a common warm state, in the next test
        assert abs(r["psnr"] - base["psnr"]) < 2.0, (name, r["psnr"], base["psnr"])
        assert 50.0 < r["psnr"] < 75.0


def test_second_half_of_the_run_from_a_common_state_agrees_within_a_tenth_of_a_db():
    """Steps 500..999 of the same 1000-step schedule in every mode, each from the SAME state (codebook, decoder, Adam moments and step counts after 500
    captured steps, quality.snapshot): past the plateau the dynamics contract, so the modes' final states are comparable at the north_star's
    resolution -- 1 bit, 0.1 dB."""
    from nerf_signature_amd import quality
    D = 32
    stage = quality.watermark_stage("hotdog")
    first = quality.train(stage, 500, "graphed", iters=1000)
    acc_mid = quality.test_bitacc(stage, 100)[0]
    snap = quality.snapshot(stage, first["optimizer"])
    print(f"\ncommon state after 500 captured steps: bit acc {acc_mid:.4f}, loss_w {first['loss_watermark']:.4f}")
    assert acc_mid > 0.9 and not first["overflowed"]
    del stage, first
    torch.cuda.empty_cache()
    recs = {}
    _run_modes(recs, 500, iters=1000, start=500, resume=snap)
    expected = _expected_adam_steps(D, 1000)
    base = recs["graphed"]
    for name, r in recs.items():
        assert r["log"][0][0] == 0 and r["log"][0][2] < 0.25, (name, r["log"][0])     # its first step (iteration 500) already decodes: resumed, not re-initialised
        assert r["adam_steps"] == expected and not r["overflowed"], name
        assert abs(r["bit_acc"] - base["bit_acc"]) <= 1.0 / 32 and r["bit_acc"] >= 1.0 - 1.0 / 32, (name, r["bit_acc"])
        assert abs(r["psnr"] - base["psnr"]) < 0.1, (name, r["psnr"], base["psnr"])                      # dB (measured: equal to 1e-3 dB)
        assert abs(r["last_lr"] - 1e-2 * 0.1 ** (999 / 1000)) < 1e-9, (name, r["last_lr"])


@pytest.mark.parametrize("scene", ["counter", "fern"])
def test_the_other_configs_learn_the_watermark_too(scene):
    """BASELINE configs 3 (two cascades, camera inside; 32 bits) and 5 (48 bits, 11 x 15 blocks, dt_gamma 1/128) through the captured loop with the README's
    schedule: from chance to (nearly) every bit right, no replay overflowing its buffers."""
    from nerf_signature_amd import quality
    rec = quality.run("graphed", 1000, scene=scene, n_messages=100)
    print(f"\n[{scene}] bit acc {rec['bit_acc_before_training']:.3f} -> {rec['bit_acc']:.4f}, PSNR vs clean views {rec['psnr_db']:.2f} dB, {rec['train_ms_per_step']:.3f} ms/step")
    assert 0.35 < rec["bit_acc_before_training"] < 0.65
    assert rec["bit_acc"] >= 0.98 and rec["wrong_bits_worst_message"] <= 3
    assert not rec["overflowed"] and rec["recaptures"] == 0
    assert 35.0 < rec["psnr_db"] < 75.0


@pytest.mark.parametrize("kind", ["brightness", "blurring", "rotation", "scaling"])
def test_robustness_training_through_the_device_distortion_layer(kind):
    """`--distortion` (utils_wtmk_disen.py:551-577,594,666) inside the captured step, draws regenerated every replay (on the device; the scaling factor
    on the host, selecting the capture of its width): after the README schedule the decoder reads the message from blocks distorted the same way (as
    the reference's eval_step distorts them) and from clean ones (rotation: trained on rotated blocks only, the unrotated ones are off-distribution)."""
    from nerf_signature_amd import quality
    rec = quality.run("graphed", 1000, n_messages=100, distortion=kind)
    print(f"\n[{kind}] bit acc clean blocks {rec['bit_acc']:.4f}, distorted blocks {rec['bit_acc_distorted_blocks']:.4f}, PSNR {rec['psnr_db']:.2f} dB, {rec['train_ms_per_step']:.3f} ms/step")
    assert rec["bit_acc"] >= (0.98 if kind != "rotation" else 0.95) and rec["bit_acc_distorted_blocks"] >= 0.97
    assert not rec["overflowed"] and 40.0 < rec["psnr_db"] < 75.0


def _small_model(seed=0, codebook_scale=1e-4):
    import test_gpu_render as T
    torch.manual_seed(seed)
    m, bitfield, C = T._model()
    with torch.no_grad():      # the reference's initialisation of the codebook (hash_encoding_wtmk_bit.py:69), closed-form stand-in
        for l in range(64):
            m.msg_encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(100 + l, scale=codebook_scale)))
    return m, bitfield, C


def test_two_hundred_steps_tracked_by_the_oracle_from_a_warm_state():
    """64 content rays + 32 blocks of 4x4 rays, README hyper-parameters.  Phase 1 (GPU only, 200 steps): the decoder leaves the chance plateau.
    Phase 2 (200 steps): the GPU loop and the oracle's autograd + torch.optim.Adam continue from that state -- parameters, Adam moments and step
    counts copied over -- on the same messages; then both are evaluated on 24 held-out messages the way test_bitacc does, and on the PSNR of the
    watermarked content render (a) against a photograph-like ground truth (clean render + a fixed +-0.03 pattern: the ~35 dB regime north_star's
    0.1 dB refers to) and (b) against each side's own clean render (the watermark's amplitude)."""
    import test_gpu_render as T
    from nerf_signature_amd import trainer
    from nerf_signature_amd.trainer import BIT_ACC
    warm, tracked, iters = 200, 200, 400
    m, bitfield, C = _small_model(seed=0)
    P, S = T._oracle_params(m, bitfield, C)
    bo, bd, co, cd, _ = T._data(n_content=64, block=4)
    kw = dict(dt_gamma=0.0, max_steps=1024)
    with torch.no_grad():
        clean = fr.render(co, cd, None, P, S, bg_color=1, **kw)["image"].clamp(0, 1)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": clean.cuda()}}
    rng = np.random.RandomState(1234)
    msgs = [torch.from_numpy(rng.randint(0, 2, 32).astype(np.float32)) for _ in range(warm + tracked)]
    lam = lambda it: 0.1 ** min(it / iters, 1)
    opt1 = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    sched1 = torch.optim.lr_scheduler.LambdaLR(opt1, lam)
    loop = trainer.WatermarkLoop(m, opt1, kw, lambda_w=0.005, lambda_i=1.0, lr_scheduler=sched1)
    for msg in msgs[:warm]:
        out = loop.step(data, msg)
    torch.cuda.synchronize()
    print(f"\nafter the warm phase ({warm} GPU steps): loss_w {float(out[4].detach()):.4f}")

    # ---- the common state, handed to the oracle
    dec_cpu = copy.deepcopy(m.msg_decoder).cpu()
    with torch.no_grad():
        for t_cpu, e in zip(P["cb_tables"], m.msg_encoder.embeddings):
            t_cpu.copy_(e.weight.detach().cpu())
    opt0 = torch.optim.Adam([{"params": P["cb_tables"]}, {"params": list(dec_cpu.parameters())}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    sd = opt1.state_dict()
    sd_cpu = {"state": {k: {n: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()},
              "param_groups": sd["param_groups"]}
    opt0.load_state_dict(sd_cpu)
    sched0 = torch.optim.lr_scheduler.LambdaLR(opt0, lam, last_epoch=-1)
    for _ in range(warm):
        sched0.step()
    assert abs(opt0.param_groups[0]["lr"] - opt1.param_groups[0]["lr"]) < 1e-12

    l1, l0 = [], []
    for msg in msgs[warm:]:
        out = loop.step(data, msg)
        l1.append([float(out[3].detach()), float(out[4].detach())])
        opt0.zero_grad(set_to_none=True)
        ref = fr.train_step(bo, bd, co, cd, clean, msg, P, S, dec_cpu, lambda_w=0.005, lambda_i=1.0, **kw)
        ref["loss"].backward()
        opt0.step()
        sched0.step()
        l0.append([float(ref["lossi"].detach()), float(ref["lossw"].detach())])
    l1, l0 = np.array(l1), np.array(l0)
    k = np.arange(0, tracked, 20)
    print("tracked phase, watermark loss every 20 steps\n  gpu   ", np.round(l1[k, 1], 4), "\n  oracle", np.round(l0[k, 1], 4))
    print("tracked phase, image loss every 20 steps\n  gpu   ", l1[k, 0], "\n  oracle", l0[k, 0])
    # the first tracked steps start from identical states: the step itself is compared (1e-3-class), before the trajectories can drift
    np.testing.assert_allclose(l1[:5, 1], l0[:5, 1], rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(l1[:5, 0], l0[:5, 0], rtol=5e-2, atol=1e-9)
    # ... and over the whole phase both learn at the same pace (mean loss over windows of 50 steps; per-step values depend on the step's message)
    w1, w0 = l1[:, 1].reshape(4, -1).mean(1), l0[:, 1].reshape(4, -1).mean(1)
    print("mean watermark loss per 50-step window: gpu", np.round(w1, 4), "oracle", np.round(w0, 4))
    np.testing.assert_allclose(w1, w0, rtol=0.1, atol=2e-3)
    assert w0[-1] < 0.5 * w0[0] or w0[-1] < 0.05              # the tracked phase is a phase of learning

    acc1, acc0 = BIT_ACC(), BIT_ACC()
    noise = torch.from_numpy(np.random.RandomState(7).uniform(-0.03, 0.03, clean.shape).astype(np.float32))
    photo = (clean + noise).clamp(0, 1)
    p1, p0, a1, a0 = [], [], [], []
    with torch.no_grad():
        c1 = m.render(co.cuda(), cd.cuda(), None, staged=False, bg_color=1, perturb=False, force_all_rays=True, **kw)["image"].cpu().double()
        c0 = fr.render(co, cd, None, P, S, bg_color=1, **kw)["image"].double()
        for s in range(24):
            msg = torch.from_numpy(np.random.RandomState(9000 + s).randint(0, 2, 32).astype(np.float32))
            _, _, _, d1, _, _, _ = trainer.eval_step(m, data["watermark"], msg.cuda(), kw, render_whole=False)
            blk0 = fr.render(bo, bd, msg, P, S, bg_color=1, **kw)["image"]
            d0 = dec_cpu(fr.normalize_img(blk0.clamp(0, 1).permute(0, 3, 1, 2)))
            acc1.update(d1.cpu().permute(1, 0), msg[None])
            acc0.update(d0.permute(1, 0), msg[None])
            if s < 6:
                i1 = m.render(co.cuda(), cd.cuda(), msg.cuda(), staged=False, bg_color=1, perturb=False, force_all_rays=True, **kw)["image"].cpu().double()
                i0 = fr.render(co, cd, msg, P, S, bg_color=1, **kw)["image"].double()
                p1.append(-10 * np.log10(float(((i1.clamp(0, 1) - photo) ** 2).mean())))
                p0.append(-10 * np.log10(float(((i0.clamp(0, 1) - photo) ** 2).mean())))
                a1.append(-10 * np.log10(float(((i1 - c1) ** 2).mean())))
                a0.append(-10 * np.log10(float(((i0 - c0) ** 2).mean())))
    print(f"bit accuracy over 24 held-out messages: gpu {acc1.measure():.4f} / oracle {acc0.measure():.4f}")
    print(f"PSNR vs photograph-like ground truth: gpu {np.mean(p1):.4f} dB / oracle {np.mean(p0):.4f} dB")
    print(f"PSNR of the watermarked render vs the clean render (the watermark's amplitude): gpu {np.mean(a1):.3f} dB / oracle {np.mean(a0):.3f} dB")
    assert acc0.measure() > 0.85 and acc1.measure() > 0.85                        # well off chance on both sides: "within one bit" can fail here
    assert abs(acc1.measure() - acc0.measure()) <= 1.0 / 32 + 1e-9                # one bit of 32
    assert abs(np.mean(p1) - np.mean(p0)) < 0.1                                   # dB, the regime the criterion is stated for
    # the watermark's own amplitude (an 86 dB signal: rms 5e-5): 0.3 dB = 3.5 % of it after 200 tracked steps (observed run to run: 0.02-0.10 dB -- the order of the
    # float atomics in G, which Adam with eps = 1e-15 turns into +-lr steps, and the fp16-operand MLPs against the oracle's fp32)
    assert abs(np.mean(a1) - np.mean(a0)) < 0.3


def test_the_bound_trainer_train_step_trains_the_watermark_through_its_captured_graphs():
    """The reference Trainer's loop shape (as tools/trainer_shape.py times it: zero_grad / autocast train_step / GradScaler / torch.optim.Adam / LambdaLR, a new
    device-side message and loader-style content rays every step) around the drop-in directory's bound Trainer.train_step -- the whole step replayed from the
    two captured graphs (blockgraph.StepGraph), the decoder's Adam step in the optimiser hook -- for the README schedule: the watermark is learnt like everywhere else."""
    import argparse
    import types
    from nerf_signature_amd import quality, synthetic, trainer
    stage = quality.watermark_stage("hotdog")
    model, dev, D, H, W = stage["model"], stage["device"], stage["D"], stage["H"], stage["W"]
    model.shared_gradient_step = model.auto_fix_rays = True          # what dropin/nerf/network_wtmk_tcnn.py switches on
    opt_ns = argparse.Namespace(**stage["render_kwargs"], num_rays=4096, lr=1e-2, workspace="x", fp16=True, color_space="srgb", loss_w="bce")
    me = types.SimpleNamespace(model=model, opt=opt_ns, lambda_w=0.005, lambda_i=1.0, distortion="none")
    optimizer = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: 0.1 ** min(it / 1000, 1))
    scaler = torch.amp.GradScaler("cuda")
    wm = {"rays_o_block": stage["block_o"], "rays_d_block": stage["block_d"], "images": torch.zeros(D, 1, 1, 3)}
    poses, clean, intr = stage["poses"], stage["clean"], stage["intr"]
    for k in range(1000):
        p = k % poses.shape[0]
        inds = torch.randint(0, H * W, size=[4096], device=dev).expand([1, 4096])
        o, d = synthetic.get_rays(poses[p:p + 1], intr, H, W, inds)
        data = {"watermark": wm, "content": {"rays_o": o, "rays_d": d, "images": torch.gather(clean[p:p + 1], 1, torch.stack(3 * [inds], -1))}}
        message = torch.randint(0, 2, (D,), dtype=torch.float32, device=dev)
        optimizer.zero_grad()
        with torch.autocast("cuda"):
            out = trainer.reference_trainer_train_step(me, data, message)
        scaler.scale(out[5]).backward()
        scaler.step(optimizer)
        scaler.update()
        scheduler.step()
    g = me._nsig_block_graph
    acc, _, worst = quality.test_bitacc(stage, 100)
    psnr = quality.test_image(stage)
    print(f"\n[bound train_step] captures {g.captures}, replays {g.generation}, overflows {g.overflows}; bit acc {acc:.4f} (worst message {worst} bits), PSNR {psnr:.2f} dB")
    assert g.failed is None and g.captures >= 1 and g.generation > 900
    assert acc >= 0.98 and worst <= 3 and 40.0 < psnr < 75.0
