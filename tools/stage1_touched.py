#!/usr/bin/env python
"""Which rows of the 16 base tables does one stage-1 step touch?  (the sizing of a sparse gradient exchange, SURVEY 8(e) / DESIGN section 7)

    python tools/stage1_touched.py [--rays 4096]

Per level: the rows with a non-zero gradient after one step of the bench scene, on the full occupancy grid a random field leaves behind (~620 k points) and on the
scene's own sparse grid (~125 k points: a trained scene), next to the STATIC bound -- the distinct rows the level's (res + 1)^3 grid corners hash to.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_signature_amd import synthetic
from nerf_signature_amd.stage1 import CleanNeRFNetwork, GraphedCleanLoop, live_rows

dev = torch.device("cuda")
n_rays = int(sys.argv[sys.argv.index("--rays") + 1]) if "--rays" in sys.argv else 4096


def fresh_model():
    m = CleanNeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
    with torch.no_grad():
        for l, e in enumerate(m.encoder.embeddings):
            e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
        grid = synthetic.density_grid(1.0)
        bits, _ = synthetic.pack_bits_np(grid, 10.0)
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bits))
    return m.to(dev).train()


def touched(refresh):
    torch.manual_seed(0)
    m = fresh_model()
    o, d = synthetic.content_rays("hotdog", n_rays, 0, dev)
    data = {"rays_o": o, "rays_d": d, "images": torch.rand(1, n_rays, 3, device=dev)}
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = GraphedCleanLoop(m, opt, dict(dt_gamma=0, max_steps=1024), n_rays=n_rays, update_extra_interval=16 if refresh else 0, perturb=True, capture=False,
                            fused_table_adam=False)
    loop.step(data)
    for _ in range(3):
        loop.step()
    torch.cuda.synchronize()
    pts = int(loop.count_ring[(loop.global_step - 1) % 16, 0])
    rows = [int((loop.g_tables[l] != 0).any(-1).sum()) for l in range(16)]
    return pts, rows


static = [int(live_rows(l).numel()) for l in range(16)]
out = {"rays": n_rays, "table_rows": 1 << 19, "static_live_rows": static}
for name, refresh in (("full_grid", True), ("sparse_grid", False)):
    pts, rows = touched(refresh)
    out[name] = {"points": pts, "touched_rows": rows, "fraction": [round(r / (1 << 19), 4) for r in rows]}
print(json.dumps(out))
