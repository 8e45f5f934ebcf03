#!/usr/bin/env python3
"""Counter target: the bench workload's training step issued EAGERLY on one stream, in the captured step's order (loop_step_begin .. block render ..
decoder .. content render .. backward .. optimiser + next march + table warm-up), so that every kernel runs in the cache context it has in the
replayed step -- in particular the block render's hash gather at the head of the step, behind the optimiser's 836 MiB stream and the warm-up pass.
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_step.py [steps]
tools/pmc_collect.py turns the run into per-(kernel, launch size) means."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from nerf_signature_amd import quality, rays, trainer
from nerf_signature_amd.optim import CodebookAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
stage = quality.watermark_stage("hotdog", codebook_scale=0.05)      # (bench.py's "trained-like" codebook)
model, dev, D, kw = stage["model"], stage["device"], stage["D"], stage["render_kwargs"]
sampler = rays.DeviceRaySampler(stage["poses"], stage["clean"], stage["intr"], stage["H"], stage["W"], 4096, seed=1000)
content = {k: torch.empty(1, 4096, 3, dtype=torch.float32, device=dev) for k in ("rays_o", "rays_d", "images")}
sampler.sample_into(torch.zeros(1, dtype=torch.int32, device=dev), content["rays_o"], content["rays_d"], content["images"])
data = {"watermark": {"rays_o_block": stage["block_o"], "rays_d_block": stage["block_d"]}, "content": content}
optimizer = CodebookAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
loop = trainer.GraphedWatermarkLoop(model, optimizer, kw, data, lambda_w=0.005, lambda_i=1.0, content_headroom=0.25, content_sampler=sampler)
msgs = quality.messages(D, steps + 2)
loop.prepare(msgs[0])           # sizing march, warm-up, capture (the capture itself launches nothing)
loop.side_stream = loop.plan_stream = loop.weights_stream = None
loop.content_backward_first = False
torch.cuda.synchronize()
print("PMC_STEP_BEGIN", flush=True)
for k in range(steps):
    loop.msg_all[:D].copy_(msgs[k].to(dev))
    loop.msg_all[D:2 * D].copy_(msgs[k + 1].to(dev))
    optimizer.zero_grad(set_to_none=True)
    loop._forward_backward()
    loop._optimise_and_march()
torch.cuda.synchronize()
n_block = int(loop.marched[0]["counter"][0])
print(f"PMC_STEP_END points: block render {n_block} (rows per launch {loop.marched[0]['capacity']}), content render capacity {loop.content_capacity}", flush=True)
