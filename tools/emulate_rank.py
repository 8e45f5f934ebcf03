"""Timing-only emulation of ONE rank's step in an R-rank job, on one GPU: a world-size-1 "nccl" group (NERFSIG_FORCE_EXCHANGE=1, so all
collectives of the step are really issued), this rank rendering D/R watermark blocks and updating the codebook tables of D/R bits, the
all-gathered block tensor filled up with copies of the rank's own blocks.  The numbers a step computes are NOT those of a real R-rank job
(the other ranks' blocks and partial pre-sums are missing); what is measured is the per-rank kernel work and launch structure at that
size, without inter-GPU latency.   usage: python tools/emulate_rank.py R [bench.py arguments]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = int(sys.argv[1])
os.environ.update(NERFSIG_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", NERFSIG_SHARD_OPTIMIZER=os.environ.get("NERFSIG_SHARD_OPTIMIZER", "1" if R >= 4 else "0"))
os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + sys.argv[2:]

from nerf_signature_amd import dp

_gather = dp._all_gather_into


def block_shard(D):
    if not dp.exchange_active() or D % R:
        return None
    return 0, D // R


def all_gather_into(out, local):
    n = local.shape[0]
    _gather(out[:n], local)                        # the real (one-rank) collective
    if out.shape[0] > n:                           # the other ranks' rows: copies of this rank's
        out[n:].view(out.shape[0] // n - 1, *local.shape).copy_(out[:n].unsqueeze(0).expand(out.shape[0] // n - 1, *local.shape))


dp.block_shard = block_shard
dp._all_gather_into = all_gather_into
sys.path.insert(0, ROOT)
import bench

bench.main()
