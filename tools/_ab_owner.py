#!/usr/bin/env python
"""hg_levels_plan + hg_levels_scatter alone on ray-like points (4096 rays x 152 samples at the bench scene's step): HIP-event time of the scatter, for diagnostic builds of
the slice owners (NERFSIG_LIB).  Run under rocprofv3 --kernel-trace --stats for the entries / owners split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_signature_amd import _native as nv

dev = torch.device("cuda")
torch.manual_seed(0)
R, S = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 152
o = torch.rand(R, 1, 3, device=dev) * 1.2 - 0.6
d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
t = (torch.arange(S, device=dev, dtype=torch.float32) - S / 2)[None, :, None] * 0.003383 + torch.rand(R, 1, 1, device=dev) * 0.003383
pts = (o + d * t).clamp(-0.999, 0.999).reshape(-1, 3).contiguous()
M = pts.shape[0]
stride = (M + 127) // 128 * 128
d_planes = torch.randn(16, stride, 2, device=dev) * 1e-4
rows = torch.tensor([M], dtype=torch.int32, device=dev)
plan = torch.empty(int(nv.fn("hg_levels_plan_bytes")(M)), dtype=torch.uint8, device=dev)
G = [torch.empty(1 << 19, 2, device=dev) for _ in range(16)]
Gp = nv.ptr_array(G)
def once():
    nv.call("hg_levels_plan", nv.ptr(pts), M, nv.ptr(rows), 1.0, nv.ptr(plan), nv.stream())
    nv.call("hg_levels_scatter", nv.ptr(pts), M, nv.ptr(rows), 1.0, nv.ptr(d_planes), stride, nv.ptr(plan), Gp, nv.stream())
for _ in range(3):
    once()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    nv.call("hg_levels_plan", nv.ptr(pts), M, nv.ptr(rows), 1.0, nv.ptr(plan), nv.stream())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    nv.call("hg_levels_scatter", nv.ptr(pts), M, nv.ptr(rows), 1.0, nv.ptr(d_planes), stride, nv.ptr(plan), Gp, nv.stream())
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(os.path.basename(os.environ.get("NERFSIG_LIB", "built")), "points", M, "hg_levels_scatter us: median", round(ts[len(ts) // 2], 1), "min", round(ts[0], 1), "checksum", float(sum(g.double().sum() for g in G)))
