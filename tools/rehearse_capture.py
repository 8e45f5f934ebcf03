import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["NERFSIG_FORCE_EXCHANGE"] = "1"
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
import subprocess
# usage (GPU box): python tools/rehearse_capture.py -> bench lines of the world-size-1 nccl rehearsal with the collectives between the
# captured segments (default) and inside one captured graph (NERFSIG_CAPTURE_COLLECTIVES=1)
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for cap in ("0", "1", "0", "1"):
    env = dict(os.environ, NERFSIG_CAPTURE_COLLECTIVES=cap)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not line:
        print(f"capture_collectives={cap}: FAILED rc={out.returncode}\n{out.stderr[-1500:]}")
        continue
    import json
    d = json.loads(line[0])
    print(f"capture_collectives={cap}: {d['ms_per_step']:.4f} ms/step, {d['config']['execution']}, collectives/step {d['config']['collectives_per_step']}, loss {d['config']['loss']:.6f}")
