"""Does the path LEARN a watermark?  The reference's whole run, on the synthetic stand-in of its data: `--iters` steps of the loop body
(/root/reference/nerf/utils_wtmk_disen.py:1164-1181) with the README's hyper-parameters (README.md:45: --iters 1000 --lambda_w 0.005
--lambda_i 1.0, lr 1e-2 decayed by 0.1 ** min(it / iters, 1), main_nerf_wtmk.py:110-116), then `test_bitacc` (:935-1030: random messages,
block render -> decoder -> BIT_ACC) and `test_image` (:816-933: staged full views with a random message against the clean views, PSNRMeter).

  watermark_stage(...)   the stage's inputs as bench.py builds them (scene S0, clean views of `n_poses` orbit poses, block rays), with the
                         codebook at the reference's initialisation U(-1e-4, 1e-4) (hash_encoding_wtmk_bit.py:69)
  train(stage, ...)      the steps, driven by the captured loop exactly as bench.py configures it ("graphed"), by the eager loop
                         the reference's Trainer shape gets ("eager"), with the block rays declared constant ("fixed"), or through a
                         world-size-1 RCCL group with every collective of the multi-rank step issued and the codebook optimiser in its
                         sharded form ("rccl1"); every mode draws the same content batches (rg_sample_rays) and the same messages
  test_bitacc / test_image   the two evaluation loops

Used by tests/test_gpu_convergence.py, bench.py's `quality` block and tools/converge.py."""
import time

import numpy as np
import torch

from . import blocks, rays, synthetic, trainer
from .distortion import DistortionLayer
from .trainer import BIT_ACC, PSNRMeter

README = dict(iters=1000, lambda_w=0.005, lambda_i=1.0, lr=1e-2)       # README.md:45 + main_nerf_wtmk.py:21


def watermark_stage(scene="hotdog", device="cuda", n_poses=8, n_test_poses=10, codebook_scale=1e-4, n_rays=4096, seed=0):
    from .network import NeRFNetwork
    dev = torch.device(device)
    cfg = synthetic.SCENES[scene]
    D, H, W = cfg["message_dim"], cfg["H"], cfg["W"]
    intr = (cfg["focal"], cfg["focal"], W / 2, H / 2)
    torch.manual_seed(seed)
    model = NeRFNetwork(bound=cfg["bound"], cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
    synthetic.init_model(model, scene, codebook_scale=codebook_scale)
    model.to(dev).train()
    kw = dict(dt_gamma=cfg["dt_gamma"], max_steps=1024)
    bo, bd = synthetic.block_rays(scene, dev)
    prng = np.random.RandomState(77)          # (bench.py's poses)
    mk = lambda n: torch.from_numpy(np.stack([synthetic.orbit_pose(0.6 + 0.9 * prng.rand(), 2 * np.pi * prng.rand(), cfg["radius"]) for _ in range(n)])).to(dev)
    poses, test_poses = mk(n_poses), mk(n_test_poses)
    with torch.no_grad():     # "ground truth" = the clean model's render of the same pose (provider_wtmk.py:408-416)
        clean = blocks.clean_render(model, poses, intr, H, W, kw, max_ray_batch=H * W).reshape(n_poses, H * W, 3).clamp_(0, 1).contiguous()
        clean_test = blocks.clean_render(model, test_poses, intr, H, W, kw, max_ray_batch=H * W).reshape(n_test_poses, H, W, 3).clamp_(0, 1).contiguous()
    return dict(model=model, scene=scene, device=dev, D=D, H=H, W=W, intr=intr, render_kwargs=kw, block_o=bo, block_d=bd, poses=poses, clean=clean,
                test_poses=test_poses, clean_test=clean_test, n_rays=n_rays)


def messages(D, n, seed=1234):
    """The per-step draws of utils_wtmk_disen.py:1165 from one seeded host stream (the same sequence in every mode and on every rank)."""
    rng = np.random.RandomState(seed)
    return [torch.from_numpy(rng.randint(0, 2, D).astype(np.float32)) for _ in range(n)]


def lr_lambda(iters):
    return lambda it: 0.1 ** min(it / iters, 1)       # main_nerf_wtmk.py:115


def snapshot(stage, optimizer, lr=README["lr"]):
    """The trainable state of a run on the host -- codebook, decoder, Adam moments and step counts -- in a form `train(..., resume=)` loads into
    a fresh stage under ANY mode (the captured loop keeps its learning rate and step counts in device tensors: normalised here)."""
    model = stage["model"]
    sd = optimizer.state_dict()
    groups = [{k: (lr if k == "lr" else v) for k, v in g.items() if k != "initial_lr"} for g in sd["param_groups"]]
    state = {k: {n: (v.detach().to("cpu", copy=True) if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()}
    params = {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items() if k.startswith(("msg_encoder.", "msg_decoder."))}
    return {"params": params, "optimizer": {"state": state, "param_groups": groups}}


def train(stage, steps=None, mode="graphed", lambda_w=README["lambda_w"], lambda_i=README["lambda_i"], lr=README["lr"], iters=None, distortion="none",
          msg_seed=1234, sampler_seed=1000, log_every=100, check_every=250, start=0, resume=None):
    """Runs `steps` training steps on stage['model'] in place.  Returns a record: losses sampled every `log_every` steps (host reads happen
    only there and at the capacity checks every `check_every` steps), wall time, capacity overflow, per-table Adam step counts.
    start / resume: continue a run at step `start` (messages, content batches and learning rate of steps start .. start+steps-1) from a
    `snapshot` taken there."""
    from .optim import CodebookAdam
    steps = README["iters"] if steps is None else steps
    iters = start + steps if iters is None else iters
    model, dev, D, kw = stage["model"], stage["device"], stage["D"], stage["render_kwargs"]
    H, W, n_rays = stage["H"], stage["W"], stage["n_rays"]
    msgs = messages(D, start + steps + 1, msg_seed)[start:]
    from . import dp as _dp
    world, rank = _dp.world_size(), _dp.rank()          # data-parallel: every rank draws its own content batches (pose k * world + rank, own pixel stream)
    sampler = rays.DeviceRaySampler(stage["poses"], stage["clean"], stage["intr"], H, W, n_rays, stride=world, offset=rank, seed=sampler_seed + rank)
    content = {k: torch.empty(1, n_rays, 3, dtype=torch.float32, device=dev) for k in ("rays_o", "rays_d", "images")}
    counter = torch.full((1,), start, dtype=torch.int32, device=dev)
    sampler.sample_into(counter, content["rays_o"], content["rays_d"], content["images"])
    data = {"watermark": {"rays_o_block": stage["block_o"], "rays_d_block": stage["block_d"]}, "content": content}
    extra = {} if distortion == "none" else {"distortion": distortion}
    graphed = mode in ("graphed", "fixed", "rccl1")
    if mode == "rccl1":
        from . import dp
        if not dp.exchange_active():
            raise RuntimeError("mode 'rccl1' needs the world-size-1 RCCL group: NERFSIG_FORCE_EXCHANGE=1 RANK=0 WORLD_SIZE=1 NERFSIG_SHARD_OPTIMIZER=1 + dp.init_from_env()")
    optimizer = CodebookAdam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15, **({"fused": True, "capturable": True} if graphed else {}))
    if resume is not None:
        missing, unexpected = model.load_state_dict(resume["params"], strict=False)
        if unexpected:
            raise ValueError(f"resume: unexpected keys {unexpected}")
        optimizer.load_state_dict(resume["optimizer"])
    schedule = lr_lambda(iters)
    shifted = (lambda it: schedule(it + start)) if (schedule is not None and start) else schedule
    if graphed:
        loop = trainer.GraphedWatermarkLoop(model, optimizer, kw, data, lambda_w=lambda_w, lambda_i=lambda_i, lr_lambda=schedule, content_headroom=0.25,
                                            content_sampler=sampler, fixed_blocks=True if mode == "fixed" else None, **extra)
        if start:
            loop.resume_at(start)
        one = lambda k: loop.step(msgs[k], next_message=msgs[k + 1])
    elif mode == "eager":
        sched = None if shifted is None else torch.optim.lr_scheduler.LambdaLR(optimizer, shifted)
        loop = trainer.WatermarkLoop(model, optimizer, kw, lambda_w=lambda_w, lambda_i=lambda_i, lr_scheduler=sched, side_stream=torch.cuda.Stream(), **extra)

        def one(k):
            counter.fill_(start + k + 1)  # the captured loop's opening kernel counts the replay before the step draws its batch
            sampler.sample_into(counter, content["rays_o"], content["rays_d"], content["images"])
            used_lr[0] = float(optimizer.param_groups[0]["lr"])
            return loop.step(data, msgs[k])
    else:
        raise ValueError(f"mode {mode!r}: graphed | eager | fixed | rccl1")
    log, recaptures, overflow = [], 0, False
    used_lr = [None]       # the learning rate the LAST step ran with
    torch.cuda.synchronize()
    t_prep = time.perf_counter()
    if graphed:          # sizing march, warm-up steps (restored afterwards), capture: set-up, timed apart from the steps
        loop.prepare(msgs[0])
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_prep = t0 - t_prep
    for k in range(steps):
        out = one(k)
        if log_every and (k % log_every == 0 or k == steps - 1):
            log.append((k, float(out[3].detach()), float(out[4].detach())))
        if graphed and check_every and k % check_every == check_every - 1 and k != steps - 1:
            if loop.ensure_capacity():      # a replay dropped rays that did not fit: buffers re-sized, step re-captured (the dropped rays stay dropped)
                recaptures += 1
                overflow = True
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if graphed:
        overflow = overflow or bool(loop.overflowed())
        loop.close()
    tables = model.msg_encoder.tables()
    counts = [float(optimizer.state[t]["step"]) if len(optimizer.state[t]) else 0.0 for t in tables]
    return dict(mode=mode, steps=steps, wall_s=wall, prepare_s=t_prep, ms_per_step=wall / steps * 1e3, log=log, overflowed=bool(overflow), recaptures=recaptures,
                adam_steps=counts, optimizer=optimizer, last_lr=float(optimizer.param_groups[0]["lr"]) if used_lr[0] is None else used_lr[0], loss_image=log[-1][1] if log else None,
                loss_watermark=log[-1][2] if log else None)


@torch.no_grad()
def state_checksums(stage):
    """Integer checksums of everything the watermark stage trains -- the 2D codebook tables and the decoder's parameters and buffers: the sum of each tensor's
    bit patterns (int64, wrapping).  Two runs leave the same list iff (up to a 2^-64 collision) they left the same bits; no 256 MB read-back."""
    model = stage["model"]
    named = [(f"msg_encoder.{i}", t) for i, t in enumerate(model.msg_encoder.tables())] + [(f"msg_decoder.{k}", v) for k, v in model.msg_decoder.state_dict().items()]
    out = []
    for name, t in named:
        t = t.detach().contiguous()
        bits = t.view(torch.int32) if t.dtype == torch.float32 else t.to(torch.int64)
        w = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=bits.device).view(bits.shape)      # (position-weighted: a permutation of the values changes the sum)
        out.append((name, int((bits.to(torch.int64) * w).sum().item())))
    return out


@torch.no_grad()
def trained_tensors(stage):
    """Copies (on the device: 256 MiB of tables at D = 32) of everything the watermark stage trains, by name -- for an element-wise comparison of two runs."""
    model = stage["model"]
    named = [(f"msg_encoder.{i}", t) for i, t in enumerate(model.msg_encoder.tables())] + [(f"msg_decoder.{k}", v) for k, v in model.msg_decoder.state_dict().items()]
    return [(name, t.detach().clone()) for name, t in named]


@torch.no_grad()
def test_bitacc(stage, n_messages=200, seed=4321, distortion="none"):
    """Trainer.test_bitacc (utils_wtmk_disen.py:935-1030): per item a random message, eval_step(render_whole=False) on the watermark blocks
    (the model stays in whatever mode it is in -- the reference never calls model.eval() here, :951), BIT_ACC over the items.
    Returns (mean bit accuracy, mean wrong bits per message, worst message's wrong bits)."""
    model, D, dev = stage["model"], stage["D"], stage["device"]
    acc = BIT_ACC()
    gen = torch.Generator(device="cpu").manual_seed(seed)
    wm = {"rays_o_block": stage["block_o"], "rays_d_block": stage["block_d"]}
    wrong = []
    layer = None if distortion in (None, "none") else DistortionLayer(distortion, seed)
    for _ in range(n_messages):
        message = torch.randint(0, 2, (D,), generator=gen).float().to(dev)
        _, _, _, decoded, _, _, _ = trainer.eval_step(model, wm, message, stage["render_kwargs"], render_whole=False, distortion=layer)
        acc.update(decoded.permute(1, 0), message.unsqueeze(0))
        wrong.append(round((1.0 - acc.instant_V) * D))
    return float(acc.measure()), float(np.mean(wrong)), int(np.max(wrong))


test_bitacc.__test__ = False


@torch.no_grad()
def test_image(stage, seed=9876, max_ray_batch=4096):
    """Trainer.test_image (:816-933): per test view a random message, eval_step(render_whole=True) -- the full view staged in
    max_ray_batch chunks -- against the clean view; PSNRMeter over the views.  Returns the mean PSNR in dB."""
    model, D, dev, H, W = stage["model"], stage["D"], stage["device"], stage["H"], stage["W"]
    meter = PSNRMeter()
    gen = torch.Generator(device="cpu").manual_seed(seed)
    for b in range(stage["test_poses"].shape[0]):
        message = torch.randint(0, 2, (D,), generator=gen).float().to(dev)
        r = rays.get_rays(stage["test_poses"][b:b + 1], stage["intr"], H, W, -1)
        data = {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "images": stage["clean_test"][b:b + 1], "H": H, "W": W}
        pred, _, gt, _, _, _, _ = trainer.eval_step(model, data, message, dict(stage["render_kwargs"], max_ray_batch=max_ray_batch), render_whole=True)
        meter.update(pred, gt)
    return float(meter.measure())


test_image.__test__ = False


LAST_STAGE = None      # (tools/converge.py's two-rank run compares the ranks' final models)


def run(mode="graphed", steps=None, scene="hotdog", n_messages=200, **train_kw):
    """watermark_stage + train + both evaluations -> the `quality` record of bench.py."""
    global LAST_STAGE
    stage = LAST_STAGE = watermark_stage(scene)
    before = test_bitacc(stage, min(n_messages, 50))[0]
    rec = train(stage, steps, mode, **train_kw)
    t0 = time.perf_counter()
    acc, wrong_mean, wrong_max = test_bitacc(stage, n_messages)
    distorted = None
    if train_kw.get("distortion", "none") != "none":      # the reference's eval_step distorts the evaluated blocks too (utils_wtmk_disen.py:666)
        distorted = test_bitacc(stage, n_messages, distortion=train_kw["distortion"])[0]
    psnr = test_image(stage)
    torch.cuda.synchronize()
    sel = [c for c in rec["adam_steps"]]
    return {"mode": mode, "steps": rec["steps"], "bit_acc": acc, "wrong_bits_mean": wrong_mean, "wrong_bits_worst_message": wrong_max, "psnr_db": psnr,
            **({} if distorted is None else {"bit_acc_distorted_blocks": distorted}),
            "wall_s": rec["wall_s"], "capture_s": rec["prepare_s"], "train_ms_per_step": rec["ms_per_step"], "eval_wall_s": time.perf_counter() - t0, "bit_acc_before_training": before,
            "n_messages": n_messages, "n_test_views": int(stage["test_poses"].shape[0]), "overflowed": rec["overflowed"], "recaptures": rec["recaptures"],
            "adam_steps_total": sum(sel), "adam_steps_min_max": [min(sel), max(sel)], "loss_image": rec["loss_image"], "loss_watermark": rec["loss_watermark"],
            "loss_log": [(k, round(a, 8), round(b, 5)) for k, a, b in rec["log"]],
            "hyper_parameters": {"lambda_w": train_kw.get("lambda_w", README["lambda_w"]), "lambda_i": train_kw.get("lambda_i", README["lambda_i"]),
                                 "lr": train_kw.get("lr", README["lr"]), "lr_schedule": "0.1 ** min(it / iters, 1)", "iters": train_kw.get("iters") or rec["steps"],
                                 "codebook_init": "U(-1e-4, 1e-4)", "decoder_init": "torch default (random)", "distortion": train_kw.get("distortion", "none")},
            "what": "scene S0 (random frozen field), README.md:45 hyper-parameters; bit accuracy over random messages as Trainer.test_bitacc, PSNR of the "
                    "watermarked full views against the clean views as Trainer.test_image"}
