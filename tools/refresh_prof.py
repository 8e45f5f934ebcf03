#!/usr/bin/env python
"""The device-side grid refresh by itself, for a kernel trace: python tools/refresh_prof.py [full|partial] [n]   (under rocprofv3 --kernel-trace --stats)
Runs n refreshes of one form eagerly on the bench scene's model (sparse occupancy grid); prints the wall time per refresh."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_signature_amd import fieldops as fo, synthetic
from nerf_signature_amd.gridrefresh import DeviceGridRefresh
from nerf_signature_amd.stage1 import CleanNeRFNetwork

form = sys.argv[1] if len(sys.argv) > 1 else "partial"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = CleanNeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
with torch.no_grad():
    for l, e in enumerate(m.encoder.embeddings):
        e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
    grid = synthetic.density_grid(1.0)
    bits, _ = synthetic.pack_bits_np(grid, 10.0)
    m.density_grid.copy_(torch.from_numpy(grid))
    m.density_bitfield.copy_(torch.from_numpy(bits))
m = m.cuda().train()
packed = fo.pack_weights(m.sigma_net.params, m.color_net.params)
r = DeviceGridRefresh(m, capture="--graph" in sys.argv)
ts = []
for _ in range(n + 2):
    m.iter_density = 0 if form == "full" else 16
    torch.cuda.synchronize()
    t = time.perf_counter()
    r.run(packed)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
print(f"{form} refresh: {sorted(ts[2:])[len(ts[2:]) // 2]:.3f} ms median of {n} ({'graph replay' if '--graph' in sys.argv else 'eager launches'})")
