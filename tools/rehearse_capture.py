"""World-size-1 "nccl" rehearsal of the multi-rank execution on one GPU (NERFSIG_FORCE_EXCHANGE=1: blocks "sharded" into one piece, all
three collectives of a step issued for real): the collectives BETWEEN captured segments (default) against captured INSIDE one graph
(NERFSIG_CAPTURE_COLLECTIVES=1).  usage (GPU box): python tools/rehearse_capture.py [pairs]  -> one line per run."""
import json
import os
import subprocess
import sys

root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
base = dict(os.environ, NERFSIG_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", NERFSIG_SHARD_OPTIMIZER="1")
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1"]
failed = 0
for k in range(pairs):
    for cap in modes:
        env = dict(base, NERFSIG_CAPTURE_COLLECTIVES=cap, MASTER_PORT=str(29655 + (k * 2 + int(cap)) % 40))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            failed += 1
            print(f"capture_collectives={cap}: FAILED rc={out.returncode}\n{out.stderr[-1500:]}", flush=True)
            continue
        d = json.loads(line[0])
        drain = d["config"].get("watchdog_drain_before_capture") or {}
        coll = (d["config"].get("collective_us") or {}).get("eager_pass_one_stream", {})
        print(f"run {k:3d} capture_collectives={cap}: {d['ms_per_step']:.4f} ms/step, {d['config']['execution']}, collectives/step {d['config']['collectives_per_step']}, loss {d['config']['loss']:.6f}; "
              f"watchdog drain: {drain.get('how')} ({drain.get('seconds', 0):.2f} s); collectives (eager pass, us): " + ", ".join(f"{n} {v['mean']:.0f}" for n, v in coll.items()), flush=True)
print(f"{failed} failed run(s)")
