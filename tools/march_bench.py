#!/usr/bin/env python
"""Times the ray-marching entry points alone on the bench rays (block and content) with HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import _native as nv
from nerf_signature_amd import raymarching as rm
from nerf_signature_amd import synthetic

dev = torch.device("cuda")
grid = synthetic.density_grid(1.0)
bits, _ = synthetic.pack_bits_np(grid)
bf = torch.from_numpy(bits).to(dev)
aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)


def timeit(fn, reps=30):
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


for name, (o, d) in {"block": synthetic.block_rays("hotdog", dev), "content": synthetic.content_rays("hotdog", 4096, 0, dev)}.items():
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    N = o.shape[0]
    nears, fars = rm.near_far_from_aabb(o, d, aabb, 0.2)
    counts = torch.empty(N, dtype=torch.int32, device=dev)
    t_rec = torch.empty(N * 1024, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    ctr = torch.zeros(2, dtype=torch.int32, device=dev)
    s = nv.stream()
    count = lambda: nv.call("rm_march_train_count", nv.ptr(o), nv.ptr(d), nv.ptr(bf), 1.0, 0.0, 1024, N, 1, 128, nv.ptr(nears), nv.ptr(fars), None, nv.ptr(counts), nv.ptr(t_rec), s)
    scan = lambda: nv.call("rm_march_train_scan", nv.ptr(counts), N, nv.ptr(rays), nv.ptr(ctr), s)
    count()
    scan()
    M = int(ctr[0].item())
    xyzs, dirs, deltas = (torch.empty(M, 3, device=dev), torch.empty(M, 3, device=dev), torch.empty(M, 2, device=dev))
    write = lambda: nv.call("rm_march_train_write", nv.ptr(o), nv.ptr(d), 1.0, 0.0, 1024, N, 1, 128, M, nv.ptr(nears), None, nv.ptr(t_rec), nv.ptr(rays), nv.ptr(ctr), nv.ptr(xyzs), nv.ptr(dirs), nv.ptr(deltas), s)
    sig = torch.rand(M, device=dev)
    rgb = torch.rand(M, 3, device=dev)
    ws, dep, img = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    write()
    comp = lambda: nv.call("rm_composite_train_fwd", nv.ptr(sig), nv.ptr(rgb), nv.ptr(deltas), nv.ptr(rays), M, N, 1e-4, nv.ptr(ws), nv.ptr(dep), nv.ptr(img), s)
    gs, gc = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
    gws, gimg = torch.rand(N, device=dev), torch.rand(N, 3, device=dev)
    comp()
    cbwd = lambda: nv.call("rm_composite_train_bwd", nv.ptr(gws), nv.ptr(gimg), nv.ptr(sig), nv.ptr(rgb), nv.ptr(deltas), nv.ptr(rays), nv.ptr(ws), nv.ptr(img), M, N, 1e-4, nv.ptr(gs), nv.ptr(gc), s)
    # the two-enqueue form (round 3): near/far inside the walk, prefix sum inside the write launch
    nf = lambda: nv.call("rm_near_far_from_aabb", nv.ptr(o), nv.ptr(d), nv.ptr(aabb), N, 0.2, nv.ptr(nears), nv.ptr(fars), s)
    count_nf = lambda: nv.call("rm_march_train_count_nf", nv.ptr(o), nv.ptr(d), nv.ptr(aabb), 0.2, nv.ptr(bf), 1.0, 0.0, 1024, N, 1, 128, None, nv.ptr(nears), nv.ptr(fars), nv.ptr(counts), nv.ptr(t_rec), s)
    scan_write = lambda: nv.call("rm_march_train_scan_write", nv.ptr(o), nv.ptr(d), 1.0, 0.0, 1024, N, 1, 128, M, nv.ptr(nears), None, nv.ptr(t_rec), nv.ptr(counts), nv.ptr(rays), nv.ptr(ctr), nv.ptr(xyzs), nv.ptr(dirs), nv.ptr(deltas), s)
    four = lambda: (nf(), count(), scan(), write())
    two = lambda: (count_nf(), scan_write())
    print(f"{name}: near/far {timeit(nf):.1f}us | count_nf {timeit(count_nf):.1f}us scan_write {timeit(scan_write):.1f}us | chain of four {timeit(four):.1f}us, chain of two {timeit(two):.1f}us")
    c = counts.cpu().numpy()
    print(f"{name}: N={N} M={M} max_count={c.max()} mean={c.mean():.1f} | count {timeit(count):.1f}us scan {timeit(scan):.1f}us "
          f"write {timeit(write):.1f}us composite fwd {timeit(comp):.1f}us bwd {timeit(cbwd):.1f}us")
    # how the index pass scales with the number of rays: flat = bound by one ray's dependent chain, linear = by issue rate
    parts = []
    for frac in (2, 4, 16, 64):
        n = N // frac
        fn = lambda: nv.call("rm_march_train_count", nv.ptr(o), nv.ptr(d), nv.ptr(bf), 1.0, 0.0, 1024, n, 1, 128, nv.ptr(nears), nv.ptr(fars), None, nv.ptr(counts), nv.ptr(t_rec), s)
        parts.append(f"N/{frac} {timeit(fn):.1f}us")
    print("   count pass on the first", " ".join(parts))
