#!/usr/bin/env python
"""Times the inference path: NeRFRenderer.render of a full 400x400 view in EVAL mode (the alive-ray burst loop, renderer_wtmk.py:323-377; what the
reference's test_step / GUI drive) on the bench scene, with the loop's control on the device (default) and read back every round
(NERFSIG_EVAL_LOOP=host); one whole view per call and staged in 4096-ray chunks.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_signature_amd import synthetic
from nerf_signature_amd.network import NeRFNetwork

real_stdout = os.dup(1)
os.dup2(2, 1)
dev = torch.device("cuda")
model = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1)
synthetic.init_model(model, "hotdog")
model.to(dev).eval()
H = W = 400
cfg = synthetic.SCENES["hotdog"]
pose = torch.from_numpy(synthetic.orbit_pose(1.1, 0.7, cfg["radius"]))[None].to(dev)
o, d = synthetic.get_rays(pose, (cfg["focal"], cfg["focal"], W / 2, H / 2), H, W)
msg = torch.from_numpy(np.random.RandomState(0).randint(0, 2, 32).astype(np.float32))
res, images = {}, {}
with torch.no_grad():
    for loop in ("device", "host"):
        os.environ["NERFSIG_EVAL_LOOP"] = loop
        for name, batch in (("whole_view", H * W), ("staged_4096", 4096)):
            for _ in range(2):
                out = model.render(o, d, msg, staged=True, max_ray_batch=batch, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                out = model.render(o, d, msg, staged=True, max_ray_batch=batch, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            res[f"{name}_{loop}_loop_ms"] = dt * 1e3
            images[(name, loop)] = out["image"].clone()
same = all(torch.equal(images[(n, "device")], images[(n, "host")]) for n in ("whole_view", "staged_4096"))
os.dup2(real_stdout, 1)
print(json.dumps(dict(res, what=f"eval-mode render of one {H}x{W} view of the bench scene (burst loop over the alive rays): control on the device vs survivor count read back every round",
                      identical_images=bool(same), mrays_per_s_whole_view_device=H * W / res["whole_view_device_loop_ms"] / 1e3)), flush=True)
