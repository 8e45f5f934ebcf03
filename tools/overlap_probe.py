#!/usr/bin/env python
"""How well do the codebook Adam (an HBM stream) and the training march (dependent chains, little memory traffic) share the GPU?
Times each alone and both at once on two streams."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import _native as nv
from nerf_signature_amd import raymarching as rm
from nerf_signature_amd import synthetic

dev = torch.device("cuda")
D, T = 32, 1 << 19
tabs = [torch.randn(T, 2, device=dev) * 1e-4 for _ in range(2 * D)]
m1 = [torch.zeros(T, 2, device=dev) for _ in range(2 * D)]
m2 = [torch.zeros(T, 2, device=dev) for _ in range(2 * D)]
steps = [torch.zeros((), device=dev) for _ in range(2 * D)]
G = torch.randn(T, 2, device=dev) * 1e-3
msg = torch.randint(0, 2, (D,), device=dev).float()
lr = torch.tensor(1e-2, device=dev)
scratch = torch.empty(2 * D, device=dev)
arrs = [nv.ptr_array(x) for x in (tabs, m1, m2, steps)]

grid = synthetic.density_grid(1.0)
bits, _ = synthetic.pack_bits_np(grid)
bf = torch.from_numpy(bits).to(dev)
aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)
o, d = synthetic.block_rays("hotdog", dev)
o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
N = o.shape[0]
nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
counts = torch.empty(N, dtype=torch.int32, device=dev)
t_rec = torch.empty(N * 1024, dtype=torch.float32, device=dev)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def adam(st):
    nv.call("opt_codebook_adam_sel", nv.ptr(G), *arrs, nv.ptr(msg), D, nv.ptr(lr), 0.9, 0.99, 1e-15, 1.0, nv.ptr(scratch), st.cuda_stream)


def march(st):
    nv.call("rm_march_train_count", nv.ptr(o), nv.ptr(d), nv.ptr(bf), 1.0, 0.0, 1024, N, 1, 128, nv.ptr(nears), nv.ptr(fars), None, nv.ptr(counts), nv.ptr(t_rec), st.cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
        s1.synchronize(); s2.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"adam alone {timed(lambda: adam(s1)):.1f} us (incl. one host sync per rep)")
print(f"march alone {timed(lambda: march(s2)):.1f} us")
print(f"both, two streams {timed(lambda: (adam(s1), march(s2))):.1f} us")
print(f"both, one stream {timed(lambda: (adam(s1), march(s1))):.1f} us")
