import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.update(NERFSIG_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", NERFSIG_SHARD_OPTIMIZER="1", MASTER_PORT="29871")
import torch
from nerf_signature_amd import dp, quality
os.dup2(2, 1)
dp.init_from_env()
for cap in ("0", "1"):
    os.environ["NERFSIG_CAPTURE_COLLECTIVES"] = cap
    for variant in ("full", "no_log", "no_lr", "no_check"):
        stage = quality.watermark_stage("hotdog")
        kw = dict(log_every=0 if variant == "no_log" else 100, check_every=0 if variant == "no_check" else 250)
        if variant == "no_lr":
            orig = quality.lr_lambda
            quality.lr_lambda = lambda iters: None
        rec = quality.train(stage, 300, "rccl1", **kw)
        if variant == "no_lr":
            quality.lr_lambda = orig
        print(f"capture={cap} {variant}: {rec['ms_per_step']:.3f} ms/step", file=sys.stderr, flush=True)
        del stage
torch.distributed.destroy_process_group()
