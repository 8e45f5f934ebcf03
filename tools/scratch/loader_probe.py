import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from nerf_signature_amd import synthetic, rays
dev = torch.device("cuda")
cfg = synthetic.SCENES["hotdog"]; H = W = 400
intr = (cfg["focal"], cfg["focal"], W / 2, H / 2)
poses = torch.from_numpy(np.stack([synthetic.orbit_pose(0.9, 0.3 * k, cfg["radius"]) for k in range(8)])).to(dev)
clean = torch.rand(8, H * W, 3, device=dev)
def loader(k):
    p = k % 8
    inds = torch.randint(0, H * W, size=[4096], device=dev).expand([1, 4096])
    o, d = synthetic.get_rays(poses[p:p + 1], intr, H, W, inds)
    images = torch.gather(clean[p:p + 1], 1, torch.stack(3 * [inds], -1))
    return o, d, images
def timeit(fn, n=200, sync_each=False):
    for k in range(10): fn(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        fn(k)
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("stand-in of the reference loader: async", round(timeit(loader), 3), "ms; with a sync after each", round(timeit(loader, sync_each=True), 3), "ms")
def ours(k):
    r = rays.get_rays(poses[k % 8:k % 8 + 1], intr, H, W, N=4096)
    return r["rays_o"], r["rays_d"], clean[k % 8][r["inds"][0]].unsqueeze(0)
print("device get_rays (rg_get_rays):    async", round(timeit(ours), 3), "ms; with a sync after each", round(timeit(ours, sync_each=True), 3), "ms")
# which op is slow
import torch.autograd.profiler as prof
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as p:
    for k in range(20): loader(k)
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=12))
