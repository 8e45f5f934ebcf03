"""Timing-only emulation of ONE rank's step in R-rank jobs (tools/emulate_rank.py's method: a world-size-1 "nccl" group so that every collective of the
step is really issued, this rank rendering D/R blocks and -- from R = 4 -- updating the tables of D/R bits), for R = 2, 4, 8 in BOTH execution modes
of the multi-rank step (collectives captured inside the step's graph / between captured segments: what bench.py's launcher tries first and second) and
for R = 8 with the block rays declared constant -- all in one process, one JSON line:  {"rank_of_2": {"captured": ms, "segmented": ms}, ...}.
Kernel work + launch structure of a rank, NO inter-GPU latency: not a measured multi-GPU number.   usage: python tools/emulate_ranks.py [--steps 50]
(Stage 1 needs no entry of its own here: its ranks render their own 4096 rays each and exchange the same 46.4 MiB at every R, so one rank's step is the same at R = 2, 4, 8 --
`tools/stage1_bench.py --rccl1` with NERFSIG_CAPTURE_COLLECTIVES=1 | 0, the bench line's secondary.stage1.exchange.{captured, segmented}.)"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(NERFSIG_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "50"

from nerf_signature_amd import dp

_gather = dp._all_gather_into
R = [1]


def block_shard(D):
    if not dp.exchange_active() or D % R[0]:
        return None
    return 0, D // R[0]


def all_gather_into(out, local):
    n = local.shape[0]
    _gather(out[:n], local)                        # the real (one-rank) collective
    if out.shape[0] > n:                           # the other ranks' rows: copies of this rank's
        out[n:].view(out.shape[0] // n - 1, *local.shape).copy_(out[:n].unsqueeze(0).expand(out.shape[0] // n - 1, *local.shape))


dp.block_shard = block_shard
dp._all_gather_into = all_gather_into
import bench

lines = []
bench.emit = lambda line, real_stdout: lines.append(line)
real_stdout = os.dup(1)
os.dup2(2, 1)
out = {}
for r, mode, fixed in ((2, "captured", False), (2, "segmented", False), (4, "captured", False), (4, "segmented", False), (8, "captured", False), (8, "segmented", False),
                       (8, "captured", True)):
    R[0] = r
    os.environ["NERFSIG_SHARD_OPTIMIZER"] = "1" if r >= 4 else "0"
    os.environ["NERFSIG_CAPTURE_COLLECTIVES"] = "1" if mode == "captured" else "0"
    sys.argv = [os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-secondary", "--windows", "1", "--steps", steps, "--warmup", "5"] + (["--fixed-blocks"] if fixed else [])
    args = bench.parse_args()
    del lines[:]
    try:
        bench.bench_training(args, "hotdog", real_stdout, None)
        j = lines[-1]
        c = j["config"]
        key = f"rank_of_{r}" + ("_fixed_blocks" if fixed else "")
        out.setdefault(key, {})[mode] = {"ms_per_step": j["ms_per_step"], "content_rays_per_s_x_ranks_before_xgmi_latency": r * j["value"], "collectives_per_step": c["collectives_per_step"],
                                         "codebook_optimizer": c["codebook_optimizer"].split(" (")[0], "execution": c["execution"], "points_per_step_per_rank": c["points_per_step_per_rank"]}
    except Exception as e:      # noqa: BLE001
        out.setdefault(f"rank_of_{r}" + ("_fixed_blocks" if fixed else ""), {})[mode] = {"error": repr(e)}
out["what"] = ("one rank of R on ONE GPU (tools/emulate_ranks.py): D/R of the 32 blocks, from R = 4 the tables of D/R bits, every collective of the step issued on a world-size-1 RCCL group; "
               "kernel work + launch structure of a rank, no inter-GPU latency; NOT measured multi-GPU numbers")
out["steps"] = int(steps)
import torch.distributed as dist
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
os.dup2(real_stdout, 1)
print(json.dumps(out), flush=True)
