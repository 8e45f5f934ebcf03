#!/usr/bin/env python
"""Stage-1 (clean model) training step on the bench scene: 4096 rays, all parameters trainable.

    python tools/stage1_bench.py [content|block] [--eager] [--steps K] [--no-refresh]

Default: the captured loop (stage1.GraphedCleanLoop), perturbed samples, the density grid refreshed every 16 steps.  --eager: the autograd
loop (stage1.CleanLoop) with the per-entry-point breakdown measured with HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import _native as nv
from nerf_signature_amd import synthetic
from nerf_signature_amd.stage1 import CleanLoop, CleanNeRFNetwork, GraphedCleanLoop

args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
which = args[0] if args else "content"
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 64
dev = torch.device("cuda")
m = CleanNeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
with torch.no_grad():
    for l, e in enumerate(m.encoder.embeddings):
        e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
    grid = synthetic.density_grid(1.0)
    bits, _ = synthetic.pack_bits_np(grid, 10.0)
    m.density_grid.copy_(torch.from_numpy(grid))
    m.density_bitfield.copy_(torch.from_numpy(bits))
m.to(dev).train()
if which == "block":
    o, d = synthetic.block_rays("hotdog", dev)
    o, d = o.reshape(1, -1, 3), d.reshape(1, -1, 3)
else:
    o, d = synthetic.content_rays("hotdog", 4096, 0, dev)
target = torch.rand(1, o.shape[1], 3, device=dev)
n_rays = o.shape[1]
opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, **({"fused": True} if "--eager" in flags else {}))
data = {"rays_o": o, "rays_d": d, "images": target, "perturb": False, "force_all_rays": True}

if "--eager" not in flags:
    loop = GraphedCleanLoop(m, opt, dict(dt_gamma=0, max_steps=1024), n_rays=n_rays, update_extra_interval=0 if "--no-refresh" in flags else 16, perturb=True,
                            overlap_plan="--no-overlap" not in flags)
    loop.step(data)
    for _ in range(15):
        loop.step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        loop.step()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    pts = int(loop.count_ring[(loop.global_step - 1) % 16, 0])
    print(f"stage-1 captured step ({which}): {n_rays} rays, {pts} points, capacity {loop.capacity}: {ms:.3f} ms/step = {n_rays / ms * 1e3:.3e} rays/s "
          f"(grid refresh every {loop.update_extra_interval} steps inside the timed region; loss {loop.losses(1)[0]:.4e}; recaptures {loop.recaptures})")
    sys.exit(0)

loop = CleanLoop(m, opt, dict(dt_gamma=0, max_steps=1024), update_extra_interval=10 ** 9)
loop.global_step = 1
events, orig = {}, nv.call


def timed(name, *a):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(name, *a)
    e1.record()
    events.setdefault(name, []).append((e0, e1))


for _ in range(3):
    loop.step(data)
torch.cuda.synchronize()
nv.call = timed
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
t0.record()
for _ in range(n):
    loop.step(data)
t1.record()
torch.cuda.synchronize()
nv.call = orig
pts = int(m.step_counter[(m.local_step - 1) % 16, 0])
ms = t0.elapsed_time(t1) / n
print(f"stage-1 eager step ({which}): {o.shape[1]} rays, {pts} points: {ms:.3f} ms/step = {o.shape[1] / ms * 1e3:.3e} rays/s")
for k, ev in sorted(events.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    tot = sum(a.elapsed_time(b) for a, b in ev) / n
    print(f"  {k:24s} {len(ev) / n:5.1f} launches/step {tot * 1e3:9.1f} us/step")
