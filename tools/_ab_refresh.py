#!/usr/bin/env python
"""Same-box A/B of NeRFRenderer.update_extra_state's full-grid probe: queried x fastest (built) against z fastest (the order the jitter is drawn in)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_signature_amd import synthetic
from nerf_signature_amd.stage1 import CleanNeRFNetwork
m = CleanNeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
with torch.no_grad():
    for l, e in enumerate(m.encoder.embeddings):
        e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
m = m.cuda().train()
grids = []
for mode in ("x fastest (built)", "z fastest (as it was)"):
    if mode.startswith("z"):
        orig = m._grid_blocks
        def gb(S, orig=orig):
            for c, i in orig(S):
                c2 = c.clone()      # (drops the block_dims attribute)
                yield c2, i
        m._grid_blocks = gb
    m.iter_density = 0
    m.density_grid.zero_(); m.density_bitfield.zero_()
    torch.manual_seed(0)
    m.update_extra_state(); torch.cuda.synchronize()
    grid0 = m.density_grid.clone()
    grids.append(grid0)
    ts = []
    for _ in range(5):
        m.iter_density = 1
        torch.cuda.synchronize(); t = time.perf_counter(); m.update_extra_state(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"{mode:24s} update_extra_state (full grid): {min(ts):.3f} ms (min of 5), {sorted(ts)[2]:.3f} median; grid checksum {float(grid0.double().sum()):.6f}")
print("the two orders leave the same grid, bit for bit:", torch.equal(grids[0], grids[1]))

# ---- the partial refresh (iter_density >= 16): scattered cells, sorted on (y, z, x) for the query (built) against unsorted
m._grid_blocks = type(m)._grid_blocks.__get__(m)
grids = []
for mode, thresh in (("sorted query (built)", type(m).PROBE_SORT_MIN), ("unsorted (as it was)", 1 << 40)):
    m.PROBE_SORT_MIN = thresh
    m.density_grid.zero_(); m.density_bitfield.zero_(); m.iter_density = 0
    torch.manual_seed(0)
    m.update_extra_state()
    m.iter_density = 16
    torch.manual_seed(1)
    m.update_extra_state(); torch.cuda.synchronize()
    grids.append(m.density_grid.clone())
    ts = []
    for _ in range(5):
        m.iter_density = 16
        torch.cuda.synchronize(); t = time.perf_counter(); m.update_extra_state(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    print(f"{mode:24s} update_extra_state (partial): {min(ts):.3f} ms (min of 5), {sorted(ts)[2]:.3f} median")
print("(the partial refresh's grids need not be equal: its cells are drawn with replacement and `fresh[indices] = sigma` keeps whichever duplicate lands last)", torch.equal(grids[0], grids[1]))
# the probe itself: the same jittered points (same generator state), sorted and unsorted query -> the same densities, bit for bit
n = m.grid_size ** 3 // 4
torch.manual_seed(5)
coords = torch.randint(0, m.grid_size, (2 * n, 3), device="cuda")
out = []
for thresh in (type(m).PROBE_SORT_MIN, 1 << 40):
    m.PROBE_SORT_MIN = thresh
    torch.manual_seed(7)
    out.append(m._probe_density(coords, 0, None))
print("sorted and unsorted probe return the same densities, bit for bit:", torch.equal(out[0], out[1]), "| two unsorted refreshes from the same state leave the same grid:", end=" ")
gs = []
for _ in range(2):
    m.PROBE_SORT_MIN = 1 << 40
    m.density_grid.zero_(); m.density_bitfield.zero_(); m.iter_density = 0
    torch.manual_seed(0); m.update_extra_state(); m.iter_density = 16
    torch.manual_seed(1); m.update_extra_state(); torch.cuda.synchronize(); gs.append(m.density_grid.clone())
print(torch.equal(gs[0], gs[1]))
