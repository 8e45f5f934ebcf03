#!/usr/bin/env python
"""Times the inference path: NeRFRenderer.render(staged=True) of a full 400x400 view in eval mode (renderer_wtmk.py:323-377,
utils_wtmk_disen.py test_step) on the bench scene."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_signature_amd import synthetic
from nerf_signature_amd.network import NeRFNetwork

dev = torch.device("cuda")
model = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1)
synthetic.init_model(model, "hotdog")
model.to(dev).eval()
H = W = 400
cfg = synthetic.SCENES["hotdog"]
pose = torch.from_numpy(synthetic.orbit_pose(1.1, 0.7, cfg["radius"]))[None].to(dev)
o, d = synthetic.get_rays(pose, (cfg["focal"], cfg["focal"], W / 2, H / 2), H, W)
msg = torch.from_numpy(np.random.RandomState(0).randint(0, 2, 32).astype(np.float32))
with torch.no_grad():
    for _ in range(2):
        out = model.render(o, d, msg, staged=True, max_ray_batch=4096 * 40, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = model.render(o, d, msg, staged=True, max_ray_batch=4096 * 40, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print(f"eval render {H}x{W}: {dt * 1e3:.1f} ms/frame = {H * W / dt / 1e6:.2f} Mrays/s, image mean {float(out['image'].mean()):.4f}")
