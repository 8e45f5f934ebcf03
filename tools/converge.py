"""Train the watermark stage to the end of the reference's schedule on the bench scene and evaluate it (nerf_signature_amd/quality.py).
    python tools/converge.py [graphed|eager|fixed|rccl1 ...] [--steps 1000] [--messages 200] [--lambda-w 0.005] [--distortion none]
One JSON line per mode on stdout.  `rccl1` runs through a world-size-1 RCCL group (every collective of the multi-rank step issued, the
codebook optimiser in its sharded form); it has to be the only mode of its process."""
import argparse
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)



def two_ranks():
    """`converge.py dp2 ...`: the same run as two REAL rank processes (gloo; both share this GPU -- RCCL refuses two ranks on one device): content rays and the D
    blocks sharded, rendered blocks all-gathered, one all-reduce of [G | decoder gradients], sharded codebook optimiser.  The parent never touches the GPU."""
    import socket
    import subprocess
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a if a != "dp2" else "graphed" for a in sys.argv[1:]]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NERFSIG_DIST_BACKEND="gloo",
                   NERFSIG_SHARD_OPTIMIZER=os.environ.get("NERFSIG_SHARD_OPTIMIZER", "1"), NERFSIG_CONVERGE_RANK="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    # the inner limit ends before the caller's (bench.py's secondaries: NERFSIG_SECONDARY_TIMEOUT_S) so that the ranks are stopped by THIS process, which knows them
    limit = max(20.0, float(os.environ.get("NERFSIG_SECONDARY_TIMEOUT_S", "255")) - 15.0)
    t0, rc = time.time(), 0
    while any(p.poll() is None for p in procs):
        if time.time() - t0 > limit or any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 5.0
            while time.time() < deadline and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rc = 1
            break
        time.sleep(0.1)
    raise SystemExit(rc or max(p.returncode or 0 for p in procs))


if "dp2" in sys.argv[1:] and os.environ.get("NERFSIG_CONVERGE_RANK") != "1":
    two_ranks()

ap = argparse.ArgumentParser()
ap.add_argument("modes", nargs="*", default=["graphed"])
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--iters", type=int, default=None)
ap.add_argument("--messages", type=int, default=200)
ap.add_argument("--lambda-w", type=float, default=0.005)
ap.add_argument("--lambda-i", type=float, default=1.0)
ap.add_argument("--lr", type=float, default=1e-2)
ap.add_argument("--distortion", default="none")
ap.add_argument("--scene", default="hotdog")
args = ap.parse_args()
if "rccl1" in args.modes:
    if len(args.modes) != 1:
        raise SystemExit("rccl1 must be the only mode of its process")
    os.environ.update(NERFSIG_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    os.environ.setdefault("NERFSIG_SHARD_OPTIMIZER", "1")
    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))

import torch

from nerf_signature_amd import dp, quality

sys.stdout.flush()
real_stdout = os.dup(1)
os.dup2(2, 1)          # (RCCL prints its banner to stdout)
dp.init_from_env()
in_dp = os.environ.get("NERFSIG_CONVERGE_RANK") == "1"
for mode in args.modes:
    rec = quality.run(mode, args.steps, scene=args.scene, n_messages=args.messages, lambda_w=args.lambda_w, lambda_i=args.lambda_i, lr=args.lr, iters=args.iters,
                      **({} if args.distortion == "none" else {"distortion": args.distortion}))
    if in_dp:
        import torch.distributed as dist
        rec["mode"] = "dp2 (two gloo ranks on one GPU, blocks + content rays sharded, codebook optimiser sharded)"
        # the replicated state must be the same on both ranks after the closing gather: codebook checksum and evaluation results
        probe = torch.tensor([sum(float(t.detach().double().abs().sum()) for t in quality.LAST_STAGE["model"].msg_encoder.tables()), rec["bit_acc"], rec["psnr_db"]], dtype=torch.float64)
        both = [torch.zeros_like(probe) for _ in range(2)]
        dist.all_gather(both, probe)
        rec["ranks_agree"] = bool(torch.allclose(both[0], both[1], rtol=1e-9, atol=0))
        rec["world_size"] = dist.get_world_size()
        if dist.get_rank() != 0:
            continue
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(rec), flush=True)
    os.dup2(2, 1)
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
