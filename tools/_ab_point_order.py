#!/usr/bin/env python
"""What a k-major point order inside a watermark block would buy the encoder (an experiment for LABNOTES, not a product path).

The block render's rays are 12 x 12 patches of ADJACENT pixels (provider_wtmk.py:470-494); the march stores a ray's samples contiguously (raymarching.cu:422-479), so
the 64 lanes of an encoder wave hold consecutive samples ALONG one ray.  Neighbours across rays at the same sample index are ~4 x closer than neighbours along a ray.
This times hg_encode_planes (mixed layout, the headline step's launch) on the same 1.29 M points in three orders: as marched (ray-major), k-major within each block of 144
rays (rays of a block sorted by pixel, samples interleaved), and randomly permuted (the floor of locality)."""
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
D, bh, bw = o.shape[:3]
o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
nears, fars = rm.near_far_from_aabb(o, d, model.aabb_train, 0.2)
xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, model.density_bitfield, 1, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)
M = xyzs.shape[0]
rays = rays.cpu().long()      # [N, 3]: ray index, offset, count
N = rays.shape[0]
# k-major inside each block: point (ray r, sample k) -> sorted by (block of r, k, r)
ray_of = torch.empty(M, dtype=torch.long)
k_of = torch.empty(M, dtype=torch.long)
for idx, off, cnt in rays.tolist():
    ray_of[off:off + cnt] = idx
    k_of[off:off + cnt] = torch.arange(cnt)
block_of = ray_of // (bh * bw)
key = (block_of * 2048 + k_of) * (bh * bw) + (ray_of % (bh * bw))
order_k = torch.argsort(key).to(dev)
# ... and k-major inside 8 x 8-ish sub-patches (here: halves of a block row pair), for a wave of 64 lanes = 64 neighbouring pixels
orders = {"ray-major (as marched)": None, "k-major inside a block (144 rays)": order_k, "random permutation": torch.randperm(M, device=dev)}
msg = torch.randint(0, 2, (32,)).float()
S = fo.codebook_presum(fo.select_tables(model.msg_encoder.tables(), fo.message_bits(msg)))
base = model.encoder.tables()
planes = torch.empty(nv.fn("hg_planes_bytes")(M), dtype=torch.uint8, device=dev)
for rnd in range(2):
    for name, order in orders.items():
        x = xyzs if order is None else xyzs[order].contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(8):
            if i == 2:
                e0.record()
            nv.call("hg_encode_planes_mixed", nv.ptr(x), M, None, 1.0, nv.ptr_array(base), nv.ptr(S), nv.ptr(planes), nv.stream())
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd}: M={M} {name:36s} {e0.elapsed_time(e1) / 6 * 1e3:7.1f} us", flush=True)
