#!/usr/bin/env python
"""Runs hg_encode_planes alone on the bench workload's block-render points (1.29 M), for counter collection."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import _native as nv
from nerf_signature_amd import fieldops as fo
from nerf_signature_amd import raymarching as rm
from nerf_signature_amd import synthetic
from nerf_signature_amd.network import NeRFNetwork

dev = torch.device("cuda")
model = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1)
synthetic.init_model(model, "hotdog")
model.to(dev).train()
o, d = synthetic.block_rays("hotdog", dev)
o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
nears, fars = rm.near_far_from_aabb(o, d, model.aabb_train, 0.2)
xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, model.density_bitfield, 1, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)
M = xyzs.shape[0]
msg = torch.randint(0, 2, (32,)).float()
S = fo.codebook_presum(fo.select_tables(model.msg_encoder.tables(), fo.message_bits(msg)))
base = model.encoder.tables()
planes = torch.empty(nv.fn("hg_planes_bytes")(M), dtype=torch.uint8, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(reps + 2):
    if i == 2:
        e0.record()
    nv.call("hg_encode_planes", nv.ptr(xyzs), M, 1.0, nv.ptr_array(base), nv.ptr(S), nv.ptr(planes), nv.stream())
e1.record()
torch.cuda.synchronize()
print(f"M={M} encode_planes {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
