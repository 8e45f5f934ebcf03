"""The reference's robustness-training modes on the captured step (`--distortion`, main_nerf_wtmk.py:75; Trainer.distortion_layer, utils_wtmk_disen.py:551-577):
for each of the five kinds the whole README schedule (1000 steps) through GraphedWatermarkLoop with the layer inside the step, then Trainer.test_bitacc on
clean blocks and on blocks distorted the same way (as the reference's eval_step does, :666).  One JSON line: per kind ms per step, bit accuracies, PSNR.
    python tools/distortion_bench.py [--steps 1000] [--messages 100] [kinds ...]"""
import argparse
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("kinds", nargs="*", default=["noise", "brightness", "blurring", "rotation", "scaling"])
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--messages", type=int, default=100)
args = ap.parse_args()

import torch  # noqa: E402,F401

from nerf_signature_amd import quality  # noqa: E402

real_stdout = os.dup(1)
os.dup2(2, 1)
out = {"what": "the README schedule through the captured step with the distortion layer inside it (noise / brightness / blurring: the decoder's first launch; rotation / "
               "scaling: one resampling launch in front of it, scaling with one capture per decoder input width), then test_bitacc on clean and on distorted blocks",
       "steps": args.steps, "n_messages": args.messages}
for kind in args.kinds:
    r = quality.run("graphed", args.steps, n_messages=args.messages, distortion=kind)
    out[kind] = {"ms_per_step": round(r["train_ms_per_step"], 4), "capture_s": round(r["capture_s"], 2), "bit_acc_clean_blocks": r["bit_acc"],
                 "bit_acc_distorted_blocks": r["bit_acc_distorted_blocks"], "psnr_db": round(r["psnr_db"], 2), "overflowed": r["overflowed"]}
    torch.cuda.empty_cache()
sys.stdout.flush()
os.dup2(real_stdout, 1)
print(json.dumps(out), flush=True)
