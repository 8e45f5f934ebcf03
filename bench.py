#!/usr/bin/env python
"""Benchmark of the watermark-stage training step (BASELINE.json: training rays/s @4096-ray batches, hotdog, 32-bit
message).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = the loop body of the reference's train_one_epoch (nerf/utils_wtmk_disen.py:1164-1181) on synthetic scene
S0: a fresh 32-bit message, the block render (32 blocks x 12 x 12 = 4608 rays) decoded by the
HiDDeN decoder, the content render (4096 rays), BCE + MSE loss, backward, gradient exchange (N > 1), Adam step on
the 32 selected codebook tables and the decoder.  Inputs are resident in HBM before the timed region.  Rays are
sharded by rank (each rank draws its own 4096 content rays; block rays and message are replicated): weak scaling.
Rank 0 prints ONE JSON line.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md chip table); measured copy ceiling 6.29e12
BYTES_FWD_PER_POINT = lambda D: 1024 + 64 * D + 48      # 16*8*8 base + D*8*8 codebook + 32 in + 16 out
BYTES_BWD_PER_POINT = lambda D: 128 * D + 64            # codebook RMW + (dsigma, drgb, ...)


class NativeTimer:
    """HIP-event timing of selected libnerfsig entry points, on the stream the kernels are launched on (torch's current
    stream).  Installed over nerf_signature_amd._native.call; each timed call records (duration, points)."""

    POINTS_ARG = {"hg_encode_planes": 1, "field_fwd": 2, "field_bwd": 1, "field_bwd_planned": 1, "hg_scatter_sliced": 1, "hg_scatter_binned": 1,
                  "hg_scatter_planned": 1, "hg_scatter_plan": 1}

    def __init__(self, nv):
        self.nv, self.orig, self.enabled = nv, nv.call, False
        self.events = {k: [] for k in self.POINTS_ARG}
        nv.call = self

    def __call__(self, name, *args):
        if not (self.enabled and name in self.events):
            return self.orig(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.orig(name, *args)
        e1.record()
        self.events[name].append((e0, e1, int(args[self.POINTS_ARG[name]])))

    def stats(self, name, min_points=0):
        """(mean seconds per launch, launches, mean points per launch) over launches with more than min_points rows."""
        ev = [(a.elapsed_time(b) * 1e-3, m) for a, b, m in self.events[name] if m > min_points]
        if not ev:
            return 0.0, 0, 0.0
        return float(np.mean([t for t, _ in ev])), len(ev), float(np.mean([m for _, m in ev]))


def cpu_baseline(model, D):
    """The oracle (CPU restatement of the reference path, reference-faithful op sequence for the encoders) timed on
    this box's host cores on a bounded sample of the same workload: 32 blocks of 8x8 rays + 2048 content rays
    (sized for roughly 10-30 s of CPU work).  Thread count: torch intra-op threads, capped at 32 -- the tensor ops of
    this sample are too small to scale further, and oversubscribing a 256-thread host made it 50x slower."""
    from nerf_signature_amd import synthetic
    from oracle import field_ref as fr
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    bo, bd = synthetic.block_rays("hotdog")
    bo, bd = bo[:, :8, :8].contiguous(), bd[:, :8, :8].contiguous()
    co, cd = synthetic.content_rays("hotdog", 2048, seed=0)
    gt = torch.rand(1, 2048, 3)
    msg = torch.randint(0, 2, (D,)).float()
    P = {"bound": 1.0, "faithful": True, "base_tables": [e.weight.detach().cpu() for e in model.encoder.embeddings],
         "cb_tables": [e.weight.detach().cpu().clone().requires_grad_(True) for e in model.msg_encoder.embeddings],
         "sigma_params": model.sigma_net.params.detach().cpu(), "color_params": model.color_net.params.detach().cpu()}
    S = {"bound": 1.0, "cascade": 1, "grid_size": 128, "density_bitfield": model.density_bitfield.cpu().numpy(),
         "aabb": np.array([-1, -1, -1, 1, 1, 1], np.float32), "min_near": 0.2, "density_scale": 1}
    dec = copy.deepcopy(model.msg_decoder).cpu()
    n_rays = bo.shape[0] * 64 + 2048
    t0 = time.perf_counter()
    out = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec, dt_gamma=0.0, max_steps=1024)
    out["loss"].backward()
    dt = time.perf_counter() - t0
    pts = out["block"]["n_points"] + out["content"]["n_points"]
    return {"value": n_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 train step (fwd+bwd, no optimiser) on {n_rays} rays = 32 blocks of 8x8 + 2048 content rays, {pts} points, {dt:.1f} s",
            "points_per_s": pts / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="issue the content render on the main stream instead of overlapping it with the block render / decoder")
    ap.add_argument("--no-fused-adam", action="store_true", help="torch's multi-tensor instead of its fused Adam kernel for the decoder parameters")
    ap.add_argument("--no-channels-last", action="store_true", help="keep the decoder in NCHW memory format")
    ap.add_argument("--no-graph", action="store_true", help="run the loop body eagerly instead of replaying the captured hipGraph")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: RCCL prints its version banner to stdout when the process group starts, so
    # everything until the result goes to stderr's descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from nerf_signature_amd import dp, fieldops, synthetic, trainer
    from nerf_signature_amd.network import NeRFNetwork

    rank, world, local_rank = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (modulo: lets a 1-GPU box exercise the N>1 code path with gloo)
    torch.cuda.set_device(dev)
    D, scene = 32, "hotdog"
    cfg = synthetic.SCENES[scene]

    torch.manual_seed(0)
    model = NeRFNetwork(bound=cfg["bound"], cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
    synthetic.init_model(model, scene)
    model.to(dev).train()
    if not args.no_channels_last and os.environ.get("NERFSIG_DECODER", "") == "torch":   # only the stock-operator decoder benefits
        model.msg_decoder.to(memory_format=torch.channels_last)   # MIOpen's kernels are NHWC: saves the layout changes around every conv
    render_kwargs = dict(dt_gamma=cfg["dt_gamma"], max_steps=1024)

    bo, bd = synthetic.block_rays(scene, dev)
    co, cd = synthetic.content_rays(scene, args.rays, seed=rank, device=dev)
    with torch.no_grad():  # "ground truth" = the clean model's render of the same rays (nerf/provider_wtmk.py:415)
        gt = model.render(co, cd, None, staged=False, bg_color=1, perturb=False, force_all_rays=True, **render_kwargs)["image"]
    data = {"watermark": {"rays_o_block": bo, "rays_d_block": bd}, "content": {"rays_o": co, "rays_d": cd, "images": gt}}
    from nerf_signature_amd.optim import CodebookAdam
    # main_nerf_wtmk.py:110: Adam(get_params(lr), betas=(0.9, 0.99), eps=1e-15) -- same semantics, the codebook update fused
    optimizer = CodebookAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, **({} if args.no_graph else ({"capturable": True} if args.no_fused_adam else {"fused": True, "capturable": True})))
    if args.no_graph:
        loop = trainer.WatermarkLoop(model, optimizer, render_kwargs, side_stream=None if args.no_overlap else torch.cuda.Stream())
    else:
        loop = trainer.GraphedWatermarkLoop(model, optimizer, render_kwargs, data, overlap_content=not args.no_overlap)

    from nerf_signature_amd import _native as nv
    timer = NativeTimer(nv)
    msg_rng = np.random.RandomState(1234)   # same stream on every rank: the message is replicated

    draw = lambda: torch.from_numpy(msg_rng.randint(0, 2, D).astype(np.float32))   # fresh message per step (:1165), host side
    upcoming = [draw()]

    def one_step():
        # the captured loop is told the next step's message one step early (the same sequence of draws, looked ahead by one):
        # its optimiser kernel then leaves that message's pre-summed codebook behind (GraphedWatermarkLoop, presum_in_adam)
        msg = upcoming.pop()
        upcoming.append(draw())
        return loop.step(data, msg) if args.no_graph else loop.step(msg, next_message=upcoming[0])

    for _ in range(args.warmup):
        one_step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = args.no_graph          # a replayed graph runs no Python: kernels are timed in the eager pass below
    t0 = time.perf_counter()
    trace = os.environ.get("NERFSIG_BENCH_TRACE")          # diagnostics only: synchronises every step
    for i in range(args.steps):
        out = one_step()
        if trace and rank == 0 and i % 5 == 4:
            print(f"[trace] step {args.warmup + i + 1}: loss_image {float(out[3]):.3e} loss_watermark {float(out[4]):.4f}", file=sys.stderr)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    overflow = None if args.no_graph else bool(loop.overflowed())
    loss_value = float(out[5].detach())
    loss_parts = (float(out[3].detach()), float(out[4].detach()))

    if not args.no_graph:
        # the same kernels, launched eagerly so that HIP events can bracket them (live, same process, same inputs)
        timer.enabled = True
        for _ in range(5):
            optimizer.zero_grad(set_to_none=True)
            loop._forward_backward()
            loop.exchange(loop.sink.G)
            loop._optimise_and_march()
        torch.cuda.synchronize()
        timer.enabled = False

    if args.no_graph:
        n_block, n_content = int(model.step_counter[(model.local_step - 2) % 16, 0]), int(model.step_counter[(model.local_step - 1) % 16, 0])
        if not args.no_overlap:      # counters are written in issue order: the overlapped step issues the content render first
            n_block, n_content = n_content, n_block
    else:
        n_block, n_content = loop.point_counts()
    rays_block, rays_content = bo.shape[0] * bo.shape[1] * bo.shape[2], args.rays
    rays_per_step = (rays_block + rays_content) * world
    # the dominant kernel: the hash-gather encoder, on the block render (the launch with the most points)
    big = max(n_block, n_content) // 2
    enc_s, enc_n, enc_rows = timer.stats("hg_encode_planes", big)
    mlp_s, _, _ = timer.stats("field_fwd", big)
    bwd_s, _, _ = timer.stats("field_bwd_planned", big)          # the block render goes through the planned binned route
    sct_s, _, _ = timer.stats("hg_scatter_planned", big)
    plan_s, _, _ = timer.stats("hg_scatter_plan", big)           # (beside the forward pass: not on the critical path)
    if bwd_s == 0.0:
        bwd_s, _, _ = timer.stats("field_bwd", big)
        sct_s, _, _ = timer.stats("hg_scatter_binned", big)
        if sct_s == 0.0:
            sct_s, _, _ = timer.stats("hg_scatter_sliced", big)
    pts_real = float(max(n_block, n_content))
    gather_bytes = 1024 + 64 * D
    achieved = pts_real * gather_bytes / enc_s if enc_s > 0 else 0.0
    fwd_total = enc_s + mlp_s
    achieved_fwd = pts_real * BYTES_FWD_PER_POINT(D) / fwd_total if fwd_total > 0 else 0.0

    if rank == 0:
        line = {
            "metric": "training rays/sec @4096 rays (hotdog, 32-bit msg)",
            "value": rays_per_step * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (hash gather, compositing) + split-bf16 MFMA with f32 accumulate (MLPs)",
            "data": "synthetic",
            "config": {
                "workload": "Blender/hotdog-like synthetic scene S0, --wtmk_tcnn, 4096 content rays + 32x12x12 block rays per rank per step, 32-bit msg, 1xMI355X HIP raymarch+hash+MLP",
                "rays_per_step_per_rank": rays_block + rays_content, "content_rays": rays_content, "block_rays": rays_block,
                "points_per_step_per_rank": n_block + n_content, "samples_per_ray_block": n_block / rays_block,
                "samples_per_ray_content": n_content / rays_content, "content_rays_per_s": rays_content * world * args.steps / elapsed,
                "message_dim": D, "parallelism": f"dp{world}", "optimizer": "Adam(betas=(0.9,0.99), eps=1e-15): torch semantics, codebook update fused (opt_codebook_adam)",
                "grad_exchange_bytes_per_step": loop.exchange.bytes_per_step,
                "execution": "eager" if args.no_graph else ("hipGraph replay (forward+backward | RCCL exchange | optimiser)" if dp.exchange_active()
                                                            else "hipGraph replay (one capture of forward+backward+optimiser)"),
                "capacity_overflow": overflow,
                "loss": loss_value, "loss_image": loss_parts[0], "loss_watermark": loss_parts[1],
            },
            "roofline": {
                "kernel": "k_encode_planes (16-level hash gather + pre-summed codebook gather, forward) on the block render",
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
                "traffic": None,
                "launches": enc_n, "avg_launch_s": enc_s, "points_per_launch": pts_real, "rows_per_launch": enc_rows,
                "algorithmic_bytes_per_point": gather_bytes,
                "note": "algorithmic bytes are SURVEY.md 8(d)'s (1024 + 64*D per point, the reference's D separate codebook gathers); "
                        "the kernel issues 1024 + 64 B/point because the D tables are pre-summed (DESIGN.md section 2)",
                "limiter": "texture-address path: PMC TA busy 84 % of the kernel, ~1.3 cycles per L1 tag lookup, 35.5 lookups per wave-level "
                           "gather (profiles/r01_pmc_encode_block_launch.txt, tools/micro/gather_rate.hip); HBM moves 412 MB per launch",
                "issued_gather_bytes_per_point": 1024 + 64,
                "frac_of_issued_bytes": (pts_real * (1024 + 64) / enc_s) / HBM_PEAK if enc_s > 0 else 0.0,
                "frac_of_measured_copy_ceiling": achieved / 6.29e12,
                "forward_encoder_plus_mlp": {"avg_s": fwd_total, "achieved_GBps": achieved_fwd / 1e9, "frac": achieved_fwd / HBM_PEAK,
                                             "algorithmic_bytes_per_point": BYTES_FWD_PER_POINT(D)},
                "backward_mlp_plus_scatter": {"avg_s": bwd_s + sct_s, "k_field_bwd_s": bwd_s, "scatter_s": sct_s, "plan_s_off_path": plan_s,
                                              "achieved_GBps": (pts_real * BYTES_BWD_PER_POINT(D) / (bwd_s + sct_s) / 1e9) if bwd_s + sct_s > 0 else 0.0,
                                              "algorithmic_bytes_per_point": BYTES_BWD_PER_POINT(D)},
            },
        }
        # the MLP kernel against the matrix-core roof (north_star: "MFMA utilisation on the MLP against chip peak").  Issued work:
        # 72 v_mfma_f32_32x32x16_bf16 (32 768 FLOP each) per 32-point wave = three split-bf16 products per algorithmic one;
        # algorithmic work: SURVEY.md 8(d)'s 10 240 MAC = 20 480 FLOP per point.  PMC cross-check (profiles/r01_pmc_mfma_summary.txt):
        # SQ_INSTS_MFMA = 2.903e6 and SQ_VALU_MFMA_BUSY_CYCLES = 9.29e7 (= 32 cycles x MFMAs) per 1.29 M-point launch.
        rows = float(enc_rows) if enc_rows else pts_real
        issued_flop = rows / 32.0 * 72 * 32768
        line["roofline_mlp"] = {
            "kernel": "k_field_fwd<planes> (sigma MLP + SH + colour MLP, split-bf16 MFMA, fp32 accumulate) on the block render",
            "bound": "mfma", "achieved": (issued_flop / mlp_s / 1e12) if mlp_s > 0 else 0.0, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": (issued_flop / mlp_s / 2.5e15) if mlp_s > 0 else 0.0, "avg_launch_s": mlp_s,
            "algorithmic_TFLOPs": (pts_real * 20480 / mlp_s / 1e12) if mlp_s > 0 else 0.0,
            "note": "achieved = issued bf16 MFMA FLOP/s (3 split-bf16 MFMAs per algorithmic product keep fp32-level accuracy); "
                    "the algorithmic rate is a third of it; the kernel is co-limited by the VALU work of splitting activations into bf16 pairs",
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                # HBM bytes per launch of the same kernel on the same inputs, from a separate rocprofv3 --pmc run (profiles/)
                line["roofline"]["traffic"] = json.load(open(pmc)).get("k_encode_planes_hbm_bytes_per_launch")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, D)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
