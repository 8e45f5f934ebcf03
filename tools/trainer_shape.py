"""The training loop a user of the UNCHANGED reference CLI gets: the reference Trainer's loop body (nerf/utils_wtmk_disen.py:1164-1190) around this
repo's model -- per step a device-side torch.randint message, optimizer.zero_grad(), train_step under autocast(fp16), GradScaler scale / step /
update, LambdaLR step, three loss.item() reads -- with the optimiser main_nerf_wtmk.py:110 builds (plain torch.optim.Adam over get_params) and the
loader's per-step work (a new pose, torch.randint pixels, the reference's meshgrid-style get_rays, the ground-truth gather: provider_wtmk.py collate).
Nothing here is captured or looked ahead.   usage: python tools/trainer_shape.py [--steps 200] [--no-fp16] [--profile] [--fix-rays]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--no-fp16", action="store_true")
ap.add_argument("--profile", action="store_true")
ap.add_argument("--fix-rays", action="store_true", help="the one call INTEGRATION.md section 5 offers: model.fix_rays(block rays)")
ap.add_argument("--no-item", action="store_true", help="skip the three per-step loss.item() reads (diagnostics)")
ap.add_argument("--plain", action="store_true", help="the model as a library user gets it (nerf_signature_amd.network.NeRFNetwork): plain autograd gradients, the optimiser's own "
                                                      "loop over D dense tables, no automatic kept-planes route -- what rounds 1-3 gave a drop-in user")
ap.add_argument("--no-shared-gradient", action="store_true")
ap.add_argument("--no-gc", action="store_true", help="diagnostics: Python's cyclic garbage collector off during the timed steps")
ap.add_argument("--gc-freeze", action="store_true", help="gc.freeze() after the warm-up steps (what the drop-in model does at its first training render)")
ap.add_argument("--phases", action="store_true", help="host wall time per phase of the loop body (perf_counter, no profiler)")
ap.add_argument("--no-auto-fix", action="store_true")
ap.add_argument("--reference-operators", action="store_true", help="NERFSIG_DROPIN_OFF=train_step: the operator sequence the reference's own Trainer.train_step issues around model.render "
                                                                   "and model.msg_decoder, instead of the method the drop-in directory binds in its place (this repo's fused train_step)")
ap.add_argument("--distortion", default="none")
ap.add_argument("--reference-loader", action="store_true", help="NERFSIG_DROPIN_OFF=get_rays: the reference's own meshgrid-style get_rays (utils_wtmk.py:57-143, ~25 small launches) in the "
                                                                "loader instead of the drop-in directory's (dropin/nerf/utils_wtmk.py -> rays.get_rays: same draws, one launch)")
ap.add_argument("--evaluate", action="store_true", help="after the timed steps: Trainer.test_bitacc over 100 random messages and test_image PSNR (quality.py) -- with --steps 330 and the "
                                                        "default three windows the loop has run the reference's whole 1000-step schedule by then")
ap.add_argument("--both", action="store_true", help="after the timed windows, time one more window with the other train_step (see --reference-operators) and report it next to the first")
args = ap.parse_args()
args.fused_step = not args.reference_operators

from nerf_signature_amd import _native as nv
from nerf_signature_amd import quality, synthetic, trainer

real_stdout = os.dup(1)
os.dup2(2, 1)
stage = quality.watermark_stage("hotdog")
model, dev, D, H, W = stage["model"], stage["device"], stage["D"], stage["H"], stage["W"]
# the drop-in module's model (nerf_signature_amd/dropin/nerf/network_wtmk_tcnn.py) switches both on
model.shared_gradient_step = not (args.plain or args.no_shared_gradient)
model.auto_fix_rays = not (args.plain or args.no_auto_fix)
opt_ns = dict(stage["render_kwargs"], num_rays=4096, lr=1e-2, workspace="x", fp16=not args.no_fp16)      # vars(opt) is splatted into render()
optimizer = torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)                       # main_nerf_wtmk.py:110
scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: 0.1 ** min(it / 1000, 1))             # :115
scaler = torch.cuda.amp.GradScaler(enabled=not args.no_fp16)                                             # utils_wtmk_disen.py:495
wm = {"rays_o_block": stage["block_o"], "rays_d_block": stage["block_d"]}
if args.fix_rays:
    model.fix_rays(wm["rays_o_block"], wm["rays_d_block"], opt_ns["dt_gamma"], opt_ns["max_steps"])
poses, clean, intr = stage["poses"], stage["clean"], stage["intr"]
intr4 = tuple(float(v) for v in intr)


def loader(k):
    p = k % poses.shape[0]
    inds = torch.randint(0, H * W, size=[4096], device=dev).expand([1, 4096])                            # utils_wtmk_disen.py:105
    if args.reference_loader or not args.fused_step:
        o, d = synthetic.get_rays(poses[p:p + 1], intr, H, W, inds)                                       # the reference's meshgrid formulation (utils_wtmk.py:57-143)
    else:                                                                                                 # what dropin/nerf/utils_wtmk.py binds: rg_get_rays
        o = torch.empty(1, 4096, 3, dtype=torch.float32, device=dev)
        d = torch.empty(1, 4096, 3, dtype=torch.float32, device=dev)
        nv.call("rg_get_rays", nv.ptr(poses[p:p + 1].contiguous()), *intr4, int(H), int(W), nv.ptr(inds.contiguous()), 1, 4096, nv.ptr(o), nv.ptr(d), nv.stream())
    images = torch.gather(clean[p:p + 1], 1, torch.stack(3 * [inds], -1))                                 # provider_wtmk.py collate
    return {"watermark": wm, "content": {"rays_o": o, "rays_d": d, "images": images}}


mse = torch.nn.MSELoss(reduction="none")                                                                 # main_nerf_wtmk.py:108, the Trainer's `criterion`
layer = None
if args.distortion != "none":
    from nerf_signature_amd.distortion import DistortionLayer
    layer = DistortionLayer(args.distortion)


def reference_shaped_train_step(data, message):
    """What the UNCHANGED reference Trainer.train_step (utils_wtmk_disen.py:579-646, 3-channel images, loss_w 'bce' :441) asks of torch and of the two
    shadowed modules, operator by operator: model.render on the blocks, clamp, distortion layer, layout change, model.normalization, model.msg_decoder,
    model.render on the content rays, an element-wise MSE criterion and its mean, BCE-with-logits on 10 x decoded, the weighted sum.  Nothing of it is
    fused: the drop-in modules only see the calls into `model`."""
    from einops import rearrange
    blocks = model.render(wm["rays_o_block"], wm["rays_d_block"], message, staged=False, bg_color=1, perturb=False, force_all_rays=True, **opt_ns)["image"]
    pred_rgb = torch.clamp(blocks, min=0, max=1)
    seen = pred_rgb
    if layer is not None:
        layer.draw(tuple(pred_rgb.shape), pred_rgb.device)
        seen = layer(pred_rgb)
    decoded = model.msg_decoder(model.normalization(rearrange(seen, "b h w c -> b c h w")))
    ct = data["content"]
    content_rgb = model.render(ct["rays_o"], ct["rays_d"], message, staged=False, bg_color=1, perturb=False, force_all_rays=True, **opt_ns)["image"]
    lossi = mse(content_rgb, ct["images"]).mean()
    lossw = torch.nn.functional.binary_cross_entropy_with_logits(decoded * 10.0, message.unsqueeze(-1), reduction="mean")
    loss = 0.005 * lossw + 1.0 * lossi
    return pred_rgb, ct["images"], content_rgb, lossi, lossw, loss


import types

me = types.SimpleNamespace(model=model, opt=argparse.Namespace(**opt_ns, color_space="srgb", loss_w="bce"), lambda_w=0.005, lambda_i=1.0, distortion=args.distortion)      # (what the method reads of the reference's Trainer)
PH = {}


def mark(name, t0):
    t1 = time.perf_counter()
    if args.phases:
        PH[name] = PH.get(name, 0.0) + (t1 - t0)
    return t1


def step(k):
    t = time.perf_counter()
    data = loader(k)
    t = mark("loader", t)
    message = torch.randint(0, 2, (D,), dtype=torch.float32, device=dev)                                 # :1165
    t = mark("message", t)
    optimizer.zero_grad()
    t = mark("zero_grad", t)
    with torch.autocast("cuda", enabled=not args.no_fp16):
        if args.fused_step:      # what the drop-in directory binds as Trainer.train_step (dropin/nerf/utils_wtmk_disen.py)
            out = trainer.reference_trainer_train_step(me, data, message)
        else:
            out = reference_shaped_train_step(data, message)
    t = mark("train_step (forward)", t)
    scaler.scale(out[5]).backward()
    t = mark("backward", t)
    scaler.step(optimizer)
    t = mark("scaler.step (unscale, inf check, optimizer)", t)
    scaler.update()
    scheduler.step()
    t = mark("scaler.update + scheduler", t)
    if not args.no_item:
        r = out[5].item(), out[3].item(), out[4].item()
        mark("three .item() reads", t)
        return r
    return None


for k in range(args.warmup):
    step(k)
torch.cuda.synchronize()
if args.profile:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
PH.clear()         # (phases: the timed steps only -- the warm-up's first calls initialise libraries)
if args.no_gc:
    import gc
    gc.collect()
    gc.disable()
if args.gc_freeze:
    import gc
    gc.collect()
    gc.freeze()
ms0 = torch.cuda.memory_stats()
windows = []
for w in range(1 if args.profile else 3):
    t0 = time.perf_counter()
    for k in range(args.steps):
        last = step(args.warmup + w * args.steps + k)
    torch.cuda.synchronize()
    windows.append((time.perf_counter() - t0) / args.steps * 1e3)
el = float(np.median(windows)) * args.steps / 1e3
quality_after = None
if args.evaluate:
    acc, wrong_mean, wrong_max = quality.test_bitacc(stage, 100)
    quality_after = {"steps_trained": args.warmup + args.steps * len(windows), "bit_acc": acc, "wrong_bits_worst_message": wrong_max, "psnr_db": quality.test_image(stage)}
other = None
if args.both and not args.profile:
    args.fused_step = not args.fused_step
    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    other = (time.perf_counter() - t0) / args.steps * 1e3
    args.fused_step = not args.fused_step
if args.profile:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(30)
if args.phases:
    ms1 = torch.cuda.memory_stats()
    print("allocator over the timed steps:", {k: ms1[k] - ms0[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "allocation.all.allocated")},
          "reserved MB", ms1["reserved_bytes.all.current"] / 1e6, file=sys.stderr)
    n = args.steps * len(windows)
    print("host wall per step by phase (ms; includes waiting for the GPU where a phase synchronises):", {k: round(v / n * 1e3, 3) for k, v in PH.items()}, file=sys.stderr)
os.dup2(real_stdout, 1)
print(json.dumps({"what": "what the UNCHANGED reference CLI runs on top of the drop-in directory: the reference Trainer's loop body (utils_wtmk_disen.py:1164-1190) -- eager, autocast(fp16) + "
                          "GradScaler, plain torch.optim.Adam, loader-style rays per step, three .item() reads per step -- around this repo's model; NOT the headline path",
                  "train_step": "Trainer.train_step as the drop-in directory binds it (dropin/nerf/utils_wtmk_disen.py -> trainer.reference_trainer_train_step: this repo's fused step)" if args.fused_step
                  else "NERFSIG_DROPIN_OFF=train_step: the reference's own operator sequence around model.render / model.msg_decoder (stock clamp, normalisation, MSE, BCE)",
                  "ms_per_step": el / args.steps * 1e3, "ms_per_step_windows": [round(w, 4) for w in windows], "content_rays_per_s": 4096 * args.steps / el, "steps": args.steps, "fp16": not args.no_fp16,
                  **({} if other is None else {("ms_per_step_with_the_references_own_train_step_operators" if args.fused_step else "ms_per_step_with_the_bound_train_step"): round(other, 4)}),
                  "block_graph": (lambda g: None if g is None else {"captures": g.captures, "replays": g.generation, "failed": g.failed})(me.__dict__.get("_nsig_block_graph")),
                  **({} if quality_after is None else {"quality_after": quality_after}),
                  "loader": "the reference's own get_rays (meshgrid formulation)" if (args.reference_loader or not args.fused_step) else "get_rays as the drop-in directory binds it (dropin/nerf/utils_wtmk.py: one launch)",
                  "fix_rays": bool(args.fix_rays), "shared_gradient_step": bool(model.shared_gradient_step), "auto_fix_rays": bool(model.auto_fix_rays), "loss": last[0] if last else None, "grad_scale": float(scaler.get_scale()) if not args.no_fp16 else None}), flush=True)
