#!/usr/bin/env python
"""Times the field kernels alone on the bench workload's points (block render: ~1.29M points) with HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import fieldops as fo
from nerf_signature_amd import raymarching as rm
from nerf_signature_amd import synthetic
from nerf_signature_amd.network import NeRFNetwork

dev = torch.device("cuda")
model = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1)
synthetic.init_model(model, "hotdog")
model.to(dev).train()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else "block"
o, d = synthetic.block_rays("hotdog", dev) if which == "block" else synthetic.content_rays("hotdog", 4096, 0, dev)
o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
nears, fars = rm.near_far_from_aabb(o, d, model.aabb_train, 0.2)
xyzs, dirs, deltas, rays = rm.march_rays_train(o, d, 1.0, model.density_bitfield, 1, 128, nears, fars, None, -1, False, 128, True, 0.0, 1024)
M = xyzs.shape[0]
msg = torch.randint(0, 2, (32,)).float()
sel = fo.select_tables(model.msg_encoder.tables(), fo.message_bits(msg))
base = model.encoder.tables()
packed = model._packed()
S = fo.codebook_presum(sel)
sig, rgb, _, masks = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True)
gs, gc = torch.randn(M, device=dev), torch.randn(M, 3, device=dev)
G = torch.zeros(1 << 19, 2, device=dev)
x01 = ((xyzs + 1) / 2).contiguous()
dfeat = fo.field_backward(xyzs, 1.0, gs, gc, sig, rgb, masks, packed, G=None, want_dfeat=True)
print(f"{which}: M={M}")
print(f"  presum            {timeit(lambda: fo.codebook_presum(sel, out=S)):8.1f} us")
print(f"  field_fwd (train, fused)  {timeit(lambda: fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=False)):8.1f} us")
print(f"  field_fwd (train, planes) {timeit(lambda: fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=True)):8.1f} us")
a = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=False)
b = fo.field_forward(xyzs, dirs, 1.0, base, S, packed, want_masks=True, planes=True)
print("  fused == planes:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), torch.equal(a[3], b[3]))
print(f"  field_fwd (infer) {timeit(lambda: fo.field_forward(xyzs, dirs, 1.0, base, S, packed)):8.1f} us")
print(f"  field_fwd (clean) {timeit(lambda: fo.field_forward(xyzs, dirs, 1.0, base, None, packed)):8.1f} us")
print(f"  encode only       {timeit(lambda: fo.encode(x01, base, S)):8.1f} us")
print(f"  field_bwd +atomics{timeit(lambda: fo.field_backward(xyzs, 1.0, gs, gc, sig, rgb, masks, packed, G=G)):8.1f} us")
print(f"  field_bwd dfeat   {timeit(lambda: fo.field_backward(xyzs, 1.0, gs, gc, sig, rgb, masks, packed, G=None, want_dfeat=True)):8.1f} us")
print(f"  codebook scatter  {timeit(lambda: fo.codebook_scatter(x01, dfeat, G)):8.1f} us")
extra = [a for a in sys.argv[2:]]
rec = fo.field_backward(xyzs, 1.0, gs, gc, sig, rgb, masks, packed, want_rec=True)
print(f"  field_bwd rec     {timeit(lambda: fo.field_backward(xyzs, 1.0, gs, gc, sig, rgb, masks, packed, want_rec=True)):8.1f} us")
print(f"  sliced scatter    {timeit(lambda: fo.codebook_scatter_sliced(rec, G)):8.1f} us")
G1, G2 = torch.zeros_like(G), torch.zeros_like(G)
fo.codebook_scatter(x01, dfeat, G1)
fo.codebook_scatter_sliced(rec, G2)
print("  sliced vs point-wise scatter: rel L2 diff", float((G1 - G2).norm() / G1.norm()), "nonzero rows equal:", bool(((G1 != 0) == (G2 != 0)).all()))
