"""The CPU baseline of bench.py (oracle, both shapes) at several intra-op thread counts: which count is the fair one on this host?
usage (GPU box): python tools/cpu_baseline_threads.py 8 16 32 64"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, json, torch; sys.path.insert(0, %r); import bench; from nerf_signature_amd import synthetic; from nerf_signature_amd.network import NeRFNetwork;"
        "torch.manual_seed(0); m = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1);"
        "synthetic.init_model(m, 'hotdog'); r = bench.cpu_baseline(m, 32, threads=int(sys.argv[1])); print(json.dumps({k: r[k] for k in ('value', 'cores', 'batch_seconds')} | {'run': r['run_shape']['value']}))" % ROOT)
for n in sys.argv[1:] or ["8", "16", "32", "64"]:
    out = subprocess.run([sys.executable, "-c", code, n], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(n, line[-1] if line else out.stderr[-400:], flush=True)
