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
for mode in args.modes:
    rec = quality.run(mode, args.steps, scene=args.scene, n_messages=args.messages, lambda_w=args.lambda_w, lambda_i=args.lambda_i, lr=args.lr, iters=args.iters,
                      **({} if args.distortion == "none" else {"distortion": args.distortion}))
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(rec), flush=True)
    os.dup2(2, 1)
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
