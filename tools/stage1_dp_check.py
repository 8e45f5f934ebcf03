#!/usr/bin/env python
"""Two REAL rank processes (gloo; both share one GPU) run the captured stage-1 loop (stage1.GraphedCleanLoop) on their own rays: the packed gradient exchange
(levels 5..15 dense + the live rows of levels 0..4 + both MLPs) between captured segments, the device-side grid refresh every 4 steps, the parameter EMA, and point
buffers sized so that rank 0 ALONE comes within 10 % of its capacity -- both ranks must then grow and re-capture in the same step.  Checked by rank 0: the replicas'
parameters, EMA shadows and density grids are identical bit for bit after every phase, both ranks re-captured once, the loss fell, the bytes per step are the packed size.

    python tools/stage1_dp_check.py            # parent: starts the two ranks (it never touches the GPU itself)
"""
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parent():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NERFSIG_DIST_BACKEND="gloo", STAGE1_DP_RANK="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc, t0 = 0, time.time()
    while any(p.poll() is None for p in procs):
        if time.time() - t0 > 300 or any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            rc = 1
            break
        time.sleep(0.1)
    sys.exit(rc or max(p.returncode or 0 for p in procs))


def rank_main():
    import torch
    import torch.distributed as dist
    from nerf_signature_amd import dp, synthetic
    from nerf_signature_amd.raymarching import padded_point_count
    from nerf_signature_amd.stage1 import CleanNeRFNetwork, GraphedCleanLoop, SPARSE_EXCHANGE_LEVELS, live_rows
    rank, world, _ = dp.init_from_env()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    assert world == 2 and dp.exchange_active()

    def model():
        m = CleanNeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
        with torch.no_grad():
            for l, e in enumerate(m.encoder.embeddings):
                e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
            grid = synthetic.density_grid(1.0)
            bits, _ = synthetic.pack_bits_np(grid, 10.0)
            m.density_grid.copy_(torch.from_numpy(grid))
            m.density_bitfield.copy_(torch.from_numpy(bits))
        torch.manual_seed(0)      # (the MLPs' random initialisation: the same on both ranks)
        with torch.no_grad():
            m.sigma_net.params.copy_(torch.randn(3072) * 0.1)
            m.color_net.params.copy_(torch.randn(7168) * 0.1)
        return m.to(dev).train()

    n_rays = 1024
    # rank 0: rays through the ball (many samples); rank 1: the same number of rays, three quarters of them missing the scene
    o, d = synthetic.content_rays("hotdog", n_rays, seed=rank, device=dev)
    if rank == 1:
        d = d.clone()
        d[0, : 3 * n_rays // 4] = torch.tensor([0.0, 1.0, 0.0], device=dev)
    gt = torch.rand(1, n_rays, 3, generator=torch.Generator().manual_seed(7 + rank)).to(dev)
    data = {"rays_o": o, "rays_d": d, "images": gt}
    kw = dict(dt_gamma=0, max_steps=1024)

    def same_on_both(tensors, what):
        for i, t in enumerate(tensors):
            both = [torch.empty_like(t) for _ in range(2)]
            dist.all_gather(both, t.contiguous())
            assert torch.equal(both[0], both[1]), f"{what} {i} differs between the ranks"

    # how many points do this rank's rays march?  (a throw-away loop sized by itself; one step)
    m = model()
    probe = GraphedCleanLoop(m, torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15), kw, n_rays=n_rays, update_extra_interval=0, perturb=False, capture=False)
    probe.step(data)
    torch.cuda.synchronize()
    mine = torch.tensor([int(probe.count_ring[0, 0])], device=dev)
    counts = [torch.empty_like(mine) for _ in range(2)]
    dist.all_gather(counts, mine)
    n0, n1 = int(counts[0]), int(counts[1])
    probe.close()
    capacity = padded_point_count(int(n0 / 0.95))
    assert n1 < 0.6 * capacity < 0.9 * capacity < n0 <= capacity, (n0, n1, capacity)

    m = model()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = GraphedCleanLoop(m, opt, kw, n_rays=n_rays, update_extra_interval=4, perturb=True, capacity=capacity, ema_decay=0.95)
    losses = []
    for it in range(14):
        loop.step(data if it == 0 else None)
        if it in (3, 9, 13):
            torch.cuda.synchronize()
            same_on_both(m.trainable(), "parameter")
            same_on_both(loop.ema_parameters(), "EMA shadow")
            same_on_both([m.density_grid, m.density_bitfield], "grid tensor")
    torch.cuda.synchronize()
    losses = loop.losses()
    n_live = sum(live_rows(l).numel() for l in SPARSE_EXCHANGE_LEVELS)
    assert loop.bytes_exchanged_per_step == (11 * (1 << 19) * 2 + 2 * n_live + 3072 + 7168) * 4
    rec = torch.tensor([loop.recaptures, loop.capacity], device=dev)
    recs = [torch.empty_like(rec) for _ in range(2)]
    dist.all_gather(recs, rec)
    assert torch.equal(recs[0], recs[1]) and int(recs[0][0]) == 1 and int(recs[0][1]) > capacity, (recs, capacity)
    assert len(loop.graph.segments) == 2 and not loop.overflowed()
    assert losses[-1] < losses[0], (losses[0], losses[-1])
    if rank == 0:
        print(f"stage-1, two gloo ranks on one GPU: {n_rays} rays each, rank 0 marches {n0} points, rank 1 {n1}; capacity {capacity} -> {int(recs[0][1])} on BOTH ranks at the same "
              f"refresh (1 re-capture each); {loop.bytes_exchanged_per_step / 2 ** 20:.2f} MiB exchanged per step between two captured segments; parameters, EMA shadows, density grid "
              f"and bitfield identical on both ranks after steps 4, 10, 14; loss {losses[0]:.4e} -> {losses[-1]:.4e}; device-side refresh every 4 steps ({m.iter_density} refreshes)")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("STAGE1_DP_RANK") == "1":
        rank_main()
    else:
        parent()
