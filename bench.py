#!/usr/bin/env python
"""Benchmark of the watermark-stage training step (BASELINE.json: training rays/s @4096-ray batches, hotdog, 32-bit
message).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N rank processes (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --config counter|fern                  # secondary workloads (BASELINE.json configs 3 and 5), not the headline

One step = the loop body of the reference's train_one_epoch (nerf/utils_wtmk_disen.py:1164-1181) on synthetic scene S0
(SURVEY.md 8(d)): a NEW pose and 4096 new random pixels (rays generated on the device, rg_get_rays; ground truth gathered
from that pose's clean render), a fresh 32-bit message, the block render (32 blocks x 12 x 12 = 4608 rays of the fixed
watermark pose) decoded by the HiDDeN decoder, the content render (4096 rays), BCE + MSE, backward, gradient exchange
(N > 1), Adam on the 32 selected codebook tables and the decoder.  Inputs (poses, clean images, block rays, tables) are
resident in HBM before the timed region.

Multi-GPU: content rays shard by rank (own poses/pixels); the D blocks are split over the ranks and the
rendered blocks all-gathered in front of the replicated decoder.  `value` = content rays (the metric's "@4096 rays" basis) x
ranks / time; the block rays are counted once in `config.all_rays_per_s`.  Weak scaling.  Rank 0 prints ONE JSON line.
"""
import argparse
import copy
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md chip table); measured copy ceiling 6.29e12
L2_PEAK = 34.5e12          # B/s, aggregate L2 -> L1 rate of the eight XCDs (MI355X_MICROARCH.md "L2 (per XCD)")
MFMA_PEAK = {"f16": 2.5e15, "bf16": 2.5e15}     # dense FLOP/s
T_BYTES = (1 << 19) * 2 * 4                      # one [2^19, 2] fp32 table


# --------------------------------------------------------------------------------------------- launcher (no GPU call in here)

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--config", choices=["hotdog", "counter", "fern"], default="hotdog", help="hotdog = the headline (BASELINE configs 2/4); counter / fern = secondary lines")
    ap.add_argument("--poses", type=int, default=8, help="pre-rendered clean views the per-step poses rotate through")
    ap.add_argument("--dry-launch", action="store_true", help="start the N rank processes over gloo, run the collectives of a step on CPU tensors, no kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch intra-op threads of the CPU baseline (default: the CPUs this process may use, at most 32)")
    ap.add_argument("--no-variant", action="store_true", help="skip the fixed-blocks variant that the default N = 1 run times behind the headline")
    ap.add_argument("--dry-device-count", type=int, default=0, help="--dry-launch: the number of GPUs the node is taken to have (default: one per rank; 1 shows the clash report)")
    ap.add_argument("--launched-by", default="", help=argparse.SUPPRESS)          # set by this file's own launcher / supervisor for the rank processes it starts
    ap.add_argument("--launch-attempt", default="", help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (BASELINE configs 3 and 5, one rank of eight emulated) that the default N = 1 run "
                                                                  "times in child processes BEFORE this process touches the GPU and reports under `secondary`")
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each: `value` / `ms_per_step` come from the FIRST (the contract's K steps); "
                                                            "median / min / max over all of them are reported under `timing`")
    ap.add_argument("--no-overlap", action="store_true", help="issue the content render on the main stream instead of overlapping it with the block render / decoder")
    ap.add_argument("--no-fused-adam", action="store_true", help="torch's multi-tensor instead of its fused Adam kernel for the decoder parameters")
    ap.add_argument("--no-graph", action="store_true", help="run the loop body eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--fixed-blocks", action="store_true", help="march the (per-dataset constant) watermark-block rays once and keep their base-level feature planes and "
                                                                   "scatter plan; a step gathers only the codebook level for them (GraphedWatermarkLoop(fixed_blocks=True))")
    ap.add_argument("--fixed-rays", action="store_true", help="replay one ray set every step (round-1 behaviour; diagnostics)")
    ap.add_argument("--host-rays", action="store_true", help="hand every step's rays over from the host loop (randint + rg_get_rays + gather + copies between replays) "
                                                                "instead of drawing them inside the captured step (rg_sample_rays)")
    return ap.parse_args()


def _run_ranks(args, n, extra_env, timeout_s, capture, extra_argv=()):
    """Start n rank processes, wait for them; returns (return code, rank 0's stdout or None, report or None).  A rank that dies takes the
    others down (they would wait in a collective for ever); a run that outlasts `timeout_s` is killed the same way -- by the exact PIDs
    started here, SIGTERM first, SIGKILL for whatever ignores it.  capture: rank 0's stdout and EVERY rank's stderr go to temporary FILES
    (never pipes: nothing here can block on a wedged child) and are read once every process is gone.  report = {"first_failed_rank": the
    rank whose non-zero exit ended the attempt (None: watchdog / none), "stderr_tail": that rank's last 20 stderr lines (rank 0's when no
    rank failed by itself), "rank0_stderr_tail": ...}: on first contact with N GPUs the rank that dies need not be rank 0, and the
    ranks that are torn down because of it (SIGTERM, possibly before they printed anything) say nothing about the cause."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    f_out = tempfile.TemporaryFile(mode="w+") if capture else None
    f_errs = [tempfile.TemporaryFile(mode="w+") for _ in range(n)] if capture else None
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--launched-by", "bench"] + list(extra_argv), env=env,
                                      stdout=f_out if (capture and r == 0) else None, stderr=f_errs[r] if capture else None))
    rc, t0, first_failed = 0, time.time(), None

    def stop(ps):
        for q in ps:
            if q.poll() is None:
                q.terminate()
        deadline = time.time() + 5.0
        while time.time() < deadline and any(q.poll() is None for q in ps):
            time.sleep(0.05)
        for q in ps:
            if q.poll() is None:
                q.kill()
        for q in ps:
            try:
                q.wait(timeout=10)
            except subprocess.TimeoutExpired:      # (unkillable: stuck in the driver; its files are still safe to read)
                pass

    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc, first_failed = code, procs.index(p)
                    stop(pending)
            if pending and timeout_s and time.time() - t0 > timeout_s:
                rc = rc or 124
                stop(pending)
                break
            time.sleep(0.05)
    finally:
        stop(procs)
    text = report = None
    if capture:
        f_out.seek(0)
        text = f_out.read()
        tails = []
        for r, f in enumerate(f_errs):
            f.seek(0)
            lines = f.readlines()
            tails.append("".join(lines[-20:]))
            if r != 0 and lines:                   # (ranks other than 0 used to write straight to this process's stderr: keep them visible)
                sys.stderr.write("".join(f"[rank {r}] " + l for l in lines[-40:]))
            f.close()
        f_out.close()
        report = {"first_failed_rank": first_failed, "stderr_tail": tails[first_failed if first_failed is not None else 0], "rank0_stderr_tail": tails[0]}
    return rc, text, report


def _attempt_chain(pinned, test_hook, rehearsal):
    attempts = []
    if pinned in (None, "", "try") or test_hook:
        attempts.append(("collectives captured inside the step's graph", {"NERFSIG_CAPTURE_COLLECTIVES": "1"}, []))
    attempts.append(("default", {"NERFSIG_CAPTURE_COLLECTIVES": "0"}, []))
    if not rehearsal:
        attempts.append(("codebook optimiser replicated", {"NERFSIG_CAPTURE_COLLECTIVES": "0", "NERFSIG_SHARD_OPTIMIZER": "0"}, []))
        attempts.append(("blocks and optimiser replicated, eager launches", {"NERFSIG_CAPTURE_COLLECTIVES": "0", "NERFSIG_SHARD_OPTIMIZER": "0",
                                                                            "NERFSIG_REPLICATE_BLOCKS": "1"}, ["--no-graph"]))
    return attempts


def _failure_line(args, n, failures, t_chain, watchdog):
    return json.dumps({"metric": "training rays/sec @4096 rays (hotdog, 32-bit msg)", "value": None, "unit": "rays/s", "n_gpus": n, "steps": args.steps,
                       "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                       "config": {"workload": "no attempt of the launcher's chain produced a line", "launch_failures": failures,
                                  "chain_seconds": round(time.time() - t_chain, 1), "watchdog_s_per_attempt": watchdog}})


def supervise_rank(args):
    """Under an EXTERNAL launcher (the driver's `python -m torch.distributed.run ... bench.py --gpus N`: RANK / WORLD_SIZE are already set)
    the launcher's chain cannot restart all ranks from one place, so every rank process becomes a supervisor of its own worker: it never
    touches the GPU, starts the real rank as a CHILD process (same RANK / WORLD_SIZE; a rendezvous of its own on MASTER_PORT + 101 + 7k, not
    the launcher's store, so a failed attempt leaves nothing behind in it), watches it with the same watchdog and walks the same chain of
    modes.  The supervisors do not talk to each other: when one rank of an attempt dies the others hang in their next collective until their
    own watchdog fires, and all of them arrive at attempt k + 1 within one watchdog period, where the fresh rendezvous waits for the last.
    Rank 0's supervisor prints its worker's JSON line, or -- when no attempt succeeded -- the `"value": null` line with the reasons.
    A pinned NERFSIG_CAPTURE_COLLECTIVES=0|1 runs the worker in this process instead, as before."""
    import tempfile
    rank, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    hook = os.environ.get("NERFSIG_TEST_FAIL_CAPTURED", "")
    rehearsal = (args.dry_launch or os.environ.get("NERFSIG_DIST_BACKEND", "") == "gloo") and hook not in ("all", "hang")
    attempts = _attempt_chain(None, hook != "", rehearsal)
    watchdog = float(os.environ.get("NERFSIG_LAUNCH_WATCHDOG_S", "90"))
    base_port = int(os.environ.get("MASTER_PORT", "29500"))
    failures, t_chain, rc, text = [], time.time(), 1, None
    current = [None]

    def on_term(signum, frame):
        # the external launcher tears every rank down as soon as ONE exits non-zero: rank 0 leaves its line first, with what it knows so far
        if current[0] is not None and current[0].poll() is None:
            current[0].kill()
        if rank == 0:
            failures.append({"attempt": len(failures), "mode": "terminated by the external launcher (another rank's supervisor gave up first)", "rc": -signum, "rank": 0})
            print(_failure_line(args, n, failures, t_chain, watchdog), flush=True)
        os._exit(1)

    import signal
    signal.signal(signal.SIGTERM, on_term)
    for k, (name, env, extra) in enumerate(attempts):
        last = k == len(attempts) - 1
        port = 1024 + (base_port + 101 + 7 * k - 1024) % 60000
        child_env = {key: v for key, v in os.environ.items() if not key.startswith("TORCHELASTIC_")}
        child_env.update(env, MASTER_ADDR="127.0.0.1",
                         MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        t_attempt = time.time()
        with tempfile.TemporaryFile(mode="w+") as f_out, tempfile.TemporaryFile(mode="w+") as f_err:
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--launched-by", "supervisor", "--launch-attempt", f"{k}: {name} (supervised under an external launcher)"]
                                 + list(extra), env=child_env, stdout=f_out, stderr=f_err)
            current[0] = p
            try:
                rc = p.wait(timeout=watchdog)
            except subprocess.TimeoutExpired:
                rc = 124
                p.terminate()
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    p.kill()
                    try:
                        p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        pass
            f_out.seek(0)
            text = f_out.read()
            f_err.seek(0)
            err = "".join(f_err.readlines()[-20:])
        if rc == 0 and (rank != 0 or (text and "{" in text)):
            break
        why = "killed by the supervisor's watchdog" if rc == 124 else ("no JSON line on rank 0's stdout" if rc == 0 else "the worker exited non-zero")
        failures.append({"attempt": k, "mode": name, "rc": rc, "why": why, "seconds": round(time.time() - t_attempt, 1), "rank": rank, "rank_stderr_tail": err[-4000:]})
        rc = rc or 1
        if rank == 0 or rc not in (124,):
            print(f"[bench] rank {rank}: attempt {k} ({name}) failed: rc {rc} ({why}) after {failures[-1]['seconds']} s; last stderr lines:\n"
                  + "".join("    | " + l + "\n" for l in err.splitlines()[-20:])
                  + (f"[bench] rank {rank}: next attempt: {attempts[k + 1][0]}" if not last else f"[bench] rank {rank}: no attempt left"), file=sys.stderr, flush=True)
    if rc == 0:
        if rank == 0:
            sys.stdout.write(text)
            sys.stdout.flush()
        raise SystemExit(0)
    if rank == 0:
        print(_failure_line(args, n, failures, t_chain, watchdog), flush=True)
    else:
        time.sleep(5.0)       # (rank 0's supervisor prints the failure line; the launcher kills it the moment this process exits non-zero)
    raise SystemExit(rc)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in their
    environment), wait, exit with the worst return code.  Decided before anything touches the GPU; children are new processes, never
    an exec of one that initialised the device.  Rank 0's stdout carries the one JSON line.

    A chain of attempts, each a fresh set of ranks under a watchdog; only the first one that ends well -- exit code 0, which includes
    the ranks agreeing on the values that must be replicated (`config.ranks_agree_on_replicated_values`) -- reaches stdout and the line
    says which it was (`config.launch_attempt`).  A multi-rank run that dies or hangs would otherwise leave no scaling point at all,
    and the modes further down the chain give up speed, not correctness:
      0. the RCCL collectives captured INSIDE the step's hipGraph (one graph per step: 1.10-1.13 against 1.175-1.20 ms on a world-size-1
         nccl group with all three collectives of a step, profiles/r02_capture_collectives_world1.txt).  Rehearsable on one rank only
         here, hence the chain behind it; the one failure seen in rehearsal (ProcessGroupNCCL's watchdog thread querying the warm-up's
         events while the step was being captured) is closed by letting the watchdog drain before the capture: 12 of 12 clean since;
      1. collectives BETWEEN captured segments, blocks sharded, codebook optimiser sharded from four ranks;
      2. the same with the codebook optimiser replicated (NERFSIG_SHARD_OPTIMIZER=0: one collective less per step);
      3. blocks replicated too (NERFSIG_REPLICATE_BLOCKS=1: round 1's partitioning, one all-reduce per step), eager launches (--no-graph).
    Every attempt -- the last one too -- runs under a watchdog (NERFSIG_LAUNCH_WATCHDOG_S, default 90 s: a healthy run takes well under
    30 s), so the whole chain ends within about four watchdogs + the kills, inside the driver's 600 s.  Every failed attempt leaves ONE
    stderr block (mode, return code, the last 20 lines of rank 0's stderr).  If no attempt succeeds the launcher still prints a JSON line
    -- `"value": null` with the reasons in `config.launch_failures` -- and exits non-zero: a scaling point that failed is on record as
    failed, not missing.
    NERFSIG_CAPTURE_COLLECTIVES=0|1 pins the mode (one attempt, no chain)."""
    n = args.gpus
    backend = os.environ.get("NERFSIG_DIST_BACKEND", "")
    if not args.dry_launch and backend != "gloo":
        have = torch.cuda.device_count()          # (does not initialise the device)
        if have < n:
            raise SystemExit(f"--gpus {n} but {have} GPU(s) visible (NERFSIG_DIST_BACKEND=gloo rehearses N ranks on fewer GPUs)")
    pinned = os.environ.get("NERFSIG_CAPTURE_COLLECTIVES")
    hook = os.environ.get("NERFSIG_TEST_FAIL_CAPTURED", "")      # tests: "1" the first attempt's ranks exit with code 3; "all": every attempt's; "hang": they sleep
    test_hook = hook != ""
    rehearsal = (args.dry_launch or backend == "gloo") and hook not in ("all", "hang")
    watchdog = float(os.environ.get("NERFSIG_LAUNCH_WATCHDOG_S", "90"))
    if ((pinned in ("0", "1") or args.no_graph or rehearsal) and not test_hook):
        rc, _, _ = _run_ranks(args, n, {}, None, capture=False)
        raise SystemExit(rc)
    attempts = _attempt_chain(pinned, test_hook, rehearsal)
    rc, text, failures = 1, None, []
    t_chain = time.time()
    for k, (name, env, extra) in enumerate(attempts):
        last = k == len(attempts) - 1
        t_attempt = time.time()
        rc, text, report = _run_ranks(args, n, env, watchdog, capture=True, extra_argv=["--launch-attempt", f"{k}: {name}"] + list(extra))
        if rc == 0 and text and "{" in text:
            break
        report = report or {"first_failed_rank": None, "stderr_tail": "", "rank0_stderr_tail": ""}
        who = report["first_failed_rank"]
        why = "killed by the launcher's watchdog" if rc == 124 else ("no JSON line on rank 0's stdout" if rc == 0 else f"rank {who} exited non-zero")
        failures.append({"attempt": k, "mode": name, "rc": rc, "why": why, "seconds": round(time.time() - t_attempt, 1), "first_failed_rank": who,
                         "failed_rank_stderr_tail": report["stderr_tail"][-4000:], "rank0_stderr_tail": report["rank0_stderr_tail"][-4000:]})
        rc = rc or 1
        print(f"[bench] attempt {k} ({name}) failed: rc {rc} ({why}) after {failures[-1]['seconds']} s; last stderr lines of rank {0 if who is None else who}:\n"
              + "".join("    | " + l + "\n" for l in report["stderr_tail"].splitlines()[-20:])
              + (f"[bench] starting the ranks again: {attempts[k + 1][0]}" if not last else "[bench] no attempt left"), file=sys.stderr, flush=True)
    if text and rc == 0:
        sys.stdout.write(text)
        sys.stdout.flush()
        raise SystemExit(0)
    # every attempt failed: the point is on record as failed
    print(_failure_line(args, n, failures, t_chain, watchdog), flush=True)
    raise SystemExit(rc)


def dry_launch(args):
    """The launcher path and the step's collectives on gloo / CPU tensors (no kernels): proves `--gpus N` starts N ranks that meet."""
    from nerf_signature_amd import dp
    hook = os.environ.get("NERFSIG_TEST_FAIL_CAPTURED", "")
    if (hook == "1" and os.environ.get("NERFSIG_CAPTURE_COLLECTIVES") == "1") or hook == "all":
        print(f"[dry-launch] test hook: rank {os.environ.get('RANK')} fails on purpose ({args.launch_attempt})", file=sys.stderr)
        raise SystemExit(3)
    if hook == "hang":
        time.sleep(3600)
    rank, world, _ = dp.init_from_env(backend="gloo")
    # first-contact check: every rank binds a GPU of its own, decided before any GPU call (no GPU here: the node is taken to have one device per
    # rank of the launch, --dry-device-count overrides -- e.g. 1 to see the clash reported)
    devices = args.dry_device_count or world
    try:
        ordinal, physical = dp.assert_distinct_devices(device_count=devices)
    except RuntimeError as e:
        print(f"[dry-launch] rank {rank}: {e}", file=sys.stderr)
        raise SystemExit(5)
    ordinals = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(ordinals, physical)
    else:
        ordinals = [physical]
    D = 32
    ok = True
    shard = dp.block_shard(D)
    if world > 1:
        ok = shard == (rank * D // world, (rank + 1) * D // world)
        local = torch.full((D // world, 2, 2, 3), float(rank))
        gathered = dp.gather_blocks(local, D, shard[0])
        ok = ok and all(float(gathered[k * (D // world)].mean()) == k for k in range(world))
    lin = torch.nn.Linear(3, 2)
    for p in lin.parameters():
        p.grad = torch.ones_like(p)
    G = torch.full((4, 2), float(rank + 1))
    ex = dp.GradExchange(list(lin.parameters()), shared_scale=1.0)
    ex(G)
    ok = ok and float(G[0, 0]) == world * (world + 1) / 2
    t = torch.tensor([1.0 if ok else 0.0])
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "world_size_seen": dp.world_size(), "backend": dist.get_backend() if dist.is_initialized() else None,
                          "block_shard_rank0": shard, "collectives_ok": bool(t.item()), "device_of_rank": ordinals, "devices_distinct": len(set(ordinals)) == world, "grad_exchange_bytes_per_step": ex.bytes_per_step,
                          "capture_collectives_env": os.environ.get("NERFSIG_CAPTURE_COLLECTIVES")}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    raise SystemExit(0 if t.item() else 1)


# --------------------------------------------------------------------------------------------- measurement helpers

class NativeTimer:
    """HIP-event timing of selected libnerfsig entry points, on the stream the kernels are launched on (torch's current
    stream).  Installed over nerf_signature_amd._native.call; each timed call records (duration, points)."""

    POINTS_ARG = {"hg_encode_planes": 1, "hg_encode_codebook_plane": 1, "field_fwd": 2, "field_bwd": 1, "field_bwd_planned": 1, "hg_scatter_sliced": 1, "hg_scatter_binned": 1,
                  "hg_scatter_planned": 1, "hg_scatter_plan": 1}

    def __init__(self, nv):
        self.nv, self.orig, self.enabled = nv, nv.call, False

        self.events = {k: [] for k in self.POINTS_ARG}
        nv.call = self

    ALIAS = {"hg_encode_planes_mixed": "hg_encode_planes"}      # (the same kernel writing the mixed plane layout: same launch, same argument position)

    def __call__(self, name, *args):
        key = self.ALIAS.get(name, name)
        if not (self.enabled and key in self.events):
            return self.orig(name, *args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.orig(name, *args)
        e1.record()
        self.events[key].append((e0, e1, int(args[self.POINTS_ARG[key]])))

    def stats(self, name, min_points=0):
        """(mean seconds per launch, launches, mean points per launch) over launches with more than min_points rows."""
        ev = [(a.elapsed_time(b) * 1e-3, m) for a, b, m in self.events[name] if m > min_points]
        if not ev:
            return 0.0, 0, 0.0
        return float(np.mean([t for t, _ in ev])), len(ev), float(np.mean([m for _, m in ev]))


# DESIGN.md section 7, written in round 4 BEFORE any run with more than one RCCL rank (ms per step, lo-hi): the emulated rank-of-R step (kernel work + launch structure of one
# rank, collectives on a one-rank group) + a latency / bandwidth budget for the collectives over xGMI.  Round 5 adds the segmented figures for R = 2, 4 from the same budget.
PREDICTED_MS = {(8, "captured"): (0.65, 0.73), (8, "segmented"): (0.86, 0.94), (4, "captured"): (0.70, 0.76), (4, "segmented"): (0.87, 0.94),
                (2, "captured"): (0.87, 0.92), (2, "segmented"): (0.96, 1.03)}


def prediction(world, mode, measured_ms, rays=4096):
    p = PREDICTED_MS.get((world, mode))
    if p is None:
        return {"note": f"no prediction on record for {world} ranks, {mode}"}
    lo, hi = p
    mid = 0.5 * (lo + hi)
    return {"source": "DESIGN.md section 7 (stated before the first multi-GPU run)", "mode": mode, "ms_per_step": [lo, hi], "rays_per_s": [rays * world / hi * 1e3, rays * world / lo * 1e3],
            "measured_ms_per_step": measured_ms, "measured_over_predicted_mid": measured_ms / mid, "inside_the_predicted_band": lo <= measured_ms <= hi}


def cpu_baseline(model, D, full_step_points=None, threads=0):
    """the oracle (CPU restatement of the reference path; reference-faithful per-bit op sequence for the
    encoders) timed on this box's host cores in BOTH shapes, 1 warm-up + 3 timed batches each, forward + backward:
      run_cuda shape (occupancy-grid march, raymarching.cu:312-693 semantics): one train step = the full 4096-ray content batch +
        the 32 watermark blocks at 4x4 of their 12x12 rays (block rays subsampled 1/9 to bound the run; stated in `sample`);
      run shape (renderer_wtmk.py:125-253: 512 uniform samples per ray, every point through encoders + sigma MLP): 512 of the 4096
        content rays (1/8), MSE against the clean image, backward to the codebook.
    Threads: torch intra-op threads = the CPUs this process may run on, capped at 32 (a 256-thread host oversubscribed these small
    tensor ops 50x in round 1); os.cpu_count() is reported beside it."""
    from nerf_signature_amd import synthetic
    from oracle import field_ref as fr
    host_cpus = os.cpu_count() or 1
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus
    quota = None
    try:      # the container's CPU share (cgroup v2 cpu.max = "<quota> <period>", or "max"): threads beyond it only take turns
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        pass
    allowed = min(allowed, quota) if quota else allowed
    # one intra-op thread per CPU this process may really use, at most 32 (same box, tools/cpu_baseline_threads.py: 4 threads 1.7e3 rays/s,
    # 8: 3.0e3, 16 = the box's share: 4.4e3, 32: 2.1e3, 64: 1.1e3, 128: 1.7e2 -- the oracle's tensor ops are small)
    torch.set_num_threads(int(threads) or max(1, min(allowed, 32)))
    bo, bd = synthetic.block_rays("hotdog")
    bo, bd = bo[:, :4, :4].contiguous(), bd[:, :4, :4].contiguous()
    co, cd = synthetic.content_rays("hotdog", 4096, seed=0)
    gt = torch.rand(1, 4096, 3)
    msg = torch.randint(0, 2, (D,)).float()
    P = {"bound": 1.0, "faithful": True, "base_tables": [e.weight.detach().cpu() for e in model.encoder.embeddings],
         "cb_tables": [e.weight.detach().cpu().clone().requires_grad_(True) for e in model.msg_encoder.embeddings],
         "sigma_params": model.sigma_net.params.detach().cpu(), "color_params": model.color_net.params.detach().cpu()}
    S = {"bound": 1.0, "cascade": 1, "grid_size": 128, "density_bitfield": model.density_bitfield.cpu().numpy(),
         "aabb": np.array([-1, -1, -1, 1, 1, 1], np.float32), "min_near": 0.2, "density_scale": 1}
    dec = copy.deepcopy(model.msg_decoder).cpu()

    def zero():
        for t in P["cb_tables"]:
            t.grad = None
        dec.zero_grad(set_to_none=True)

    def step_cuda_shape():
        zero()
        out = fr.train_step(bo, bd, co, cd, gt, msg, P, S, dec, dt_gamma=0.0, max_steps=1024)
        out["loss"].backward()
        return out["block"]["n_points"] + out["content"]["n_points"]

    def step_run_shape():
        zero()
        out = fr.run_uniform(co[:, :512], cd[:, :512], msg, P, S, num_steps=512, bg_color=1)
        ((out["image"] - gt[:, :512]) ** 2).mean().backward()
        return 512 * 512

    res = {}
    for name, fn, rays in (("run_cuda", step_cuda_shape, 4096 + bo.shape[0] * 16), ("run", step_run_shape, 512)):
        pts = fn()                                     # warm-up
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            pts = fn()
            times.append(time.perf_counter() - t0)
        res[name] = {"rays_per_s": rays / float(np.mean(times)), "points_per_s": pts / float(np.mean(times)), "rays": rays, "points": pts,
                     "batch_s": [round(t, 3) for t in times]}
    a, b = res["run_cuda"], res["run"]
    same = None
    if full_step_points:      # the GPU line's basis: CONTENT rays per second of a FULL step (all 4608 block rays + 4096 content rays)
        estimate = 4096.0 / (full_step_points / a["points_per_s"])
        same = {"value": estimate, "unit": "content rays/s", "estimated": True,
                "how": f"4096 content rays / (the full step's {int(full_step_points)} points / this baseline's measured points_per_s): the oracle's time is proportional to "
                       "its point count (every point runs the same encoders + MLPs forward and backward)"}
        # ... and MEASURED once: the whole 8704-ray step (all 32 x 12 x 12 block rays + the 4096 content rays, ~1.4 M points, forward + backward), one warm-up + one
        # timed pass.  ~14 GB of autograd records on the host: skipped (the estimate stays) when the box has less than 24 GB free.
        try:
            import psutil
            free_gb = psutil.virtual_memory().available / 2 ** 30
        except Exception:
            free_gb = 0.0
        if free_gb >= 24.0:
            bo_f, bd_f = synthetic.block_rays("hotdog")

            def full():
                zero()
                o = fr.train_step(bo_f, bd_f, co, cd, gt, msg, P, S, dec, dt_gamma=0.0, max_steps=1024)
                o["loss"].backward()
                return o["block"]["n_points"] + o["content"]["n_points"]
            full()
            t0 = time.perf_counter()
            pts_full = full()
            t_full = time.perf_counter() - t0
            same = {"value": 4096.0 / t_full, "unit": "content rays/s", "estimated": False, "full_step_s": round(t_full, 3), "points": int(pts_full), "rays": 4096 + bo_f.shape[0] * bo_f.shape[1] * bo_f.shape[2],
                    "points_per_s": pts_full / t_full, "estimate_from_the_subsampled_steps": estimate,
                    "how": "one warm-up + ONE timed full step of the oracle (fr.train_step + backward): 4096 content rays + all 4608 block rays, the GPU line's step"}
        else:
            same["why_not_measured"] = f"{free_gb:.0f} GiB of host memory free (< 24)"
    return {"value": a["rays_per_s"], "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "same_basis": same,
            "comparable_with_the_gpu_line": "ONLY `points_per_s` (vs config.points_per_s) and `same_basis.value` (vs `value`): this entry's own `value` counts the 4608 rays of a step whose "
                                            "block rays are subsampled 1/9 (4096 content + 512 block rays, ~270 k points), the GPU line's `value` counts the 4096 content rays of a step "
                                            "that also renders all 4608 block rays (~1.41 M points)",
            "sample": f"run_cuda shape: 1 warm-up + 3 timed train steps (fwd+bwd, no optimiser) of {a['rays']} rays = the full 4096-ray content batch + "
                      f"32 blocks at 4x4 of their 12x12 rays (block rays subsampled 1/9), {a['points']} points, {np.mean(a['batch_s']):.1f} s per step; "
                      "same_basis: 1 warm-up + 1 timed FULL step (all block rays)",
            "points_per_s": a["points_per_s"], "host_cpus": host_cpus, "cpus_allowed": allowed, "cgroup_cpu_quota": quota, "torch_threads": torch.get_num_threads(),
            "run_shape": {"value": b["rays_per_s"], "unit": "rays/s", "points_per_s": b["points_per_s"],
                          "sample": f"run shape (512 uniform samples/ray, renderer_wtmk.py:125-253): 1 warm-up + 3 timed fwd+bwd batches of 512 of the 4096 content rays "
                                    f"(1/8), {b['points']} points, {np.mean(b['batch_s']):.1f} s per batch"},
            "batch_seconds": {"run_cuda": a["batch_s"], "run": b["batch_s"]}}


def _run_secondary(argv, limit_s, extra_env=None):
    """One secondary child in a session (process group) of its own, output into temporary files; at the limit the WHOLE group is killed -- a child that
    starts rank processes of its own (tools/converge.py dp2) must not leave them training on the GPU beside the next secondary or the headline."""
    import signal
    import tempfile
    env = dict(os.environ, NERFSIG_SECONDARY_TIMEOUT_S=str(limit_s), **(extra_env or {}))
    with tempfile.TemporaryFile(mode="w+") as f_out, tempfile.TemporaryFile(mode="w+") as f_err:
        p = subprocess.Popen([sys.executable] + list(argv), env=env, stdout=f_out, stderr=f_err, start_new_session=True)
        timed_out = False
        try:
            rc = p.wait(timeout=limit_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 10.0)):
                try:
                    os.killpg(p.pid, sig)          # (the session leader's pid is the group's id: exactly the processes started here)
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=grace)
                except subprocess.TimeoutExpired:
                    continue
            rc = p.poll() if p.poll() is not None else 124
            try:
                os.killpg(p.pid, signal.SIGKILL)   # grandchildren that outlived a leader which exited on SIGTERM
            except ProcessLookupError:
                pass
        f_out.seek(0)
        f_err.seek(0)
        return {"rc": rc if not timed_out else (rc or 124), "stdout": f_out.read(), "stderr": f_err.read(), "timed_out": timed_out}


def run_secondaries(args):
    """What else the driver's run should put on record, each timed by a fresh child process, one after the other, started and finished BEFORE this
    process initialises the GPU (a process that has touched the device must not start others on this pool, and the headline must not share
    the GPU with them); a failing child costs its own entry only:
      quality        the reference's whole run on this path -- 1000 captured steps with README.md:45's hyper-parameters, then Trainer.test_bitacc over 200
                     messages and Trainer.test_image against the clean views (tools/converge.py, nerf_signature_amd/quality.py) -> the line's `quality`
      quality_two_ranks_gloo   the same run as two real rank processes sharing this GPU over gloo (RCCL refuses two ranks on one device): the data-parallel
                     step -- blocks + content rays sharded, all-gather, one all-reduce, sharded optimiser -- trained to the end; its ms per step is gloo's, not a figure of merit
      counter, fern  BASELINE.json configs 3 and 5
      rank_emulation one rank of 2 / 4 / 8 on this GPU, in both execution modes of the multi-rank step + fixed blocks (tools/emulate_ranks.py: kernel work and
                     launch structure of a rank, every collective issued on a world-size-1 RCCL group, no xGMI latency)
      eager_reference_trainer_shape   the loop a user of the UNCHANGED reference CLI drives: the reference Trainer's loop body and its train_step's operator
                     sequence (stock clamp / normalisation / MSE / BCE around model.render and model.msg_decoder) around this repo's model, eager, autocast(fp16) +
                     GradScaler, plain torch.optim.Adam, loader-style rays and three .item() reads per step (tools/trainer_shape.py); next to it the same loop
                     around this repo's fused trainer.train_step
      eval_loop      the eval-mode burst loop (renderer_wtmk.py:335-372) on one 400x400 view, control on the device vs read back every round (tools/eval_bench.py)
      distortion_layer   the five `--distortion` kinds inside the captured step, each trained through the README schedule: ms per step, bit accuracy on clean and on
                     distorted blocks (tools/distortion_bench.py)
      stage1         the clean model's training step (all parameters trainable, stage1.GraphedCleanLoop, grid refresh every 16 steps inside the timed
                     windows): ms per step, rays/s, per-kernel times, roofline records of the 16-level table scatter and the weight-gradient reduction (tools/stage1_bench.py)
    Every child runs in a process group of its own and the whole group is killed at the limit (NERFSIG_SECONDARY_TIMEOUT_S, default 150 s each)."""
    out = {}
    k = str(min(args.steps, 50))
    jobs = (("quality", [os.path.join(ROOT, "tools", "converge.py"), "graphed", "--steps", "1000", "--messages", "200"]),
            ("quality_two_ranks_gloo", [os.path.join(ROOT, "tools", "converge.py"), "dp2", "--steps", "1000", "--messages", "100"]),
            ("counter", [os.path.abspath(__file__), "--config", "counter", "--steps", k, "--warmup", str(min(args.warmup, 5)), "--no-cpu-baseline", "--no-secondary", "--windows", "1"]),
            ("fern", [os.path.abspath(__file__), "--config", "fern", "--steps", "5", "--warmup", "1", "--no-secondary"]),
            ("rank_emulation", [os.path.join(ROOT, "tools", "emulate_ranks.py"), "--steps", k]),
            ("eager_reference_trainer_shape", [os.path.join(ROOT, "tools", "trainer_shape.py"), "--steps", "330", "--both", "--evaluate"]),
            ("eval_loop", [os.path.join(ROOT, "tools", "eval_bench.py")]),
            ("distortion_layer", [os.path.join(ROOT, "tools", "distortion_bench.py")]),
            ("stage1", [os.path.join(ROOT, "tools", "stage1_bench.py"), "content", "--json", "--steps", "64", "--windows", "5"]),
            ("stage1_counter", [os.path.join(ROOT, "tools", "stage1_bench.py"), "content", "--scene", "counter", "--json", "--steps", "64", "--windows", "3"]),      # two cascades (config-3-like S1)
            # one rank of a data-parallel stage-1 job: the same step with its gradient exchange issued for real on a world-size-1 RCCL group (kernel work + launch
            # structure of a rank, no xGMI latency), collectives inside the step's graph | between two captured segments
            ("stage1_exchange_captured", [os.path.join(ROOT, "tools", "stage1_bench.py"), "content", "--json", "--rccl1", "--steps", "64", "--windows", "3"], {"NERFSIG_CAPTURE_COLLECTIVES": "1"}),
            ("stage1_exchange_segmented", [os.path.join(ROOT, "tools", "stage1_bench.py"), "content", "--json", "--rccl1", "--steps", "64", "--windows", "3"], {"NERFSIG_CAPTURE_COLLECTIVES": "0"}))
    limit = float(os.environ.get("NERFSIG_SECONDARY_TIMEOUT_S", "150"))
    for name, argv, *job_env in jobs:
        t0 = time.time()
        try:
            r = _run_secondary(argv, limit, job_env[0] if job_env else None)
            lines = [l for l in r["stdout"].splitlines() if l.startswith("{")]
            if r["rc"] != 0 or not lines:
                out[name] = {"error": f"rc {r['rc']}" + (" (killed with its whole process group at the time limit)" if r["timed_out"] else ""), "stderr_tail": r["stderr"][-600:]}
                continue
            j = json.loads(lines[-1])
            c = j.get("config", {})
            if name in ("quality", "quality_two_ranks_gloo", "rank_emulation", "eager_reference_trainer_shape", "eval_loop", "distortion_layer", "stage1"):
                out[name] = j
            elif name == "stage1_counter":                 # -> secondary.stage1.counter
                out.setdefault("stage1", {})["counter"] = {k: j.get(k) for k in ("ms_per_step", "rays_per_s", "points_per_step", "cascades", "recaptures", "capacity_overflow", "loss_last")}
                out["stage1"]["counter"]["sparse_grid_ms_per_step"] = (j.get("sparse_grid") or {}).get("ms_per_step")
                continue
            elif name.startswith("stage1_exchange_"):      # -> secondary.stage1.exchange.{captured, segmented}
                sg = j.get("sparse_grid") or {}
                s1 = out.setdefault("stage1", {})
                if not isinstance(s1.get("exchange"), dict):
                    s1["exchange"] = {}
                s1["exchange"][name[len("stage1_exchange_"):]] = {
                    "ms_per_step": j["ms_per_step"], "points_per_step": j["points_per_step"], "sparse_grid_ms_per_step": sg.get("ms_per_step"), **(j.get("exchange") or {})}
                continue
            elif name == "counter":
                out[name] = {"ms_per_step": j["ms_per_step"], "content_rays_per_s": j["value"], "points_per_s": c.get("points_per_s"), "points_per_step": c.get("points_per_step_per_rank"),
                             "samples_per_ray_block": c.get("samples_per_ray_block"), "samples_per_ray_content": c.get("samples_per_ray_content"), "steps": j["steps"],
                             "workload": "BASELINE config 3: Mip-NeRF360/counter-like synthetic scene S1 (bound 2, two cascades, camera inside), 4096 content + 4608 block rays, 32-bit msg",
                             "parity": "tests/test_gpu_fullsize.py[counter], tests/test_gpu_ref_native.py::test_bench_workload_march_vs_reference[counter]"}
            elif name == "fern":
                out[name] = {"ms_per_image": j["ms_per_step"], "images_per_s": j["value"], "rays_per_s": c.get("rays_per_s"), "chunks": c.get("chunks"), "steps": j["steps"],
                             "workload": "BASELINE config 5: LLFF/fern-like synthetic scene S2, 1008x756 view staged in 187 chunks of 4096 rays + 48 blocks of 11x15 through the decoder, 48-bit msg",
                             "parity": "tests/test_gpu_fullsize.py::test_fern_*"}
            out[name]["child_wall_s"] = round(time.time() - t0, 1)
        except Exception as e:       # noqa: BLE001 -- a secondary figure must never take the headline down
            out[name] = {"error": repr(e)}
    return out


LINE_LIMIT = 8000          # bytes of the ONE stdout line (the driver keeps a 9 KB tail of stdout + stderr; round 5's 23.9 KB line came back unparsed)
DETAIL_FILE = "bench_detail.json"


def _r(x, digits=6):
    """Numbers at the precision a record needs (6 significant digits), everything else untouched."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    return x


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full):
    """The line a machine reads: numbers and short identifiers only.  What every number means: DESIGN.md section 6 ("The bench line"); the complete record (prose,
    per-kernel tables, every secondary child's output) goes to DETAIL_FILE."""
    c, rf = full.get("config", {}), full.get("roofline", {})
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    scene = "hotdog" if "hotdog" in str(full.get("metric", "")) else ("counter" if "counter" in str(full.get("metric", "")) else "other")
    line["config"] = {"workload": {"hotdog": "BASELINE configs[1]: Blender/hotdog-like S0, --wtmk_tcnn, 4096 content + 32x12x12 block rays, 32-bit msg",
                                   "counter": "BASELINE configs[2]: Mip-NeRF360/counter-like S1 (bound 2, two cascades), 4096 content + 32x12x12 block rays"}.get(scene, str(c.get("workload", ""))[:120]),
                      **_pick(c, "content_rays", "block_rays_total", "block_rays_this_rank", "blocks_sharded_over_ranks", "points_per_step_per_rank", "samples_per_ray_block",
                              "samples_per_ray_content", "message_dim", "parallelism", "points_per_s", "all_rays_per_s", "grad_exchange_bytes_per_step", "collectives_per_step",
                              "backend", "ranks_agree_on_replicated_values", "rays_per_s", "chunks", "bit_accuracy_random_init", "capacity_overflow", "recaptured_with_more_headroom_after_warmup", "loss", "loss_image", "loss_watermark"),
                      "execution": "eager" if str(c.get("execution", "")).startswith("eager") else "hipgraph", "deterministic_gradient": c.get("deterministic_gradient")}
    seg = re.search(r"(\d+) captured segment", str(c.get("execution", "")))
    if seg:
        line["config"]["graph_segments"] = int(seg.group(1))
    if isinstance(c.get("fixed_blocks_variant"), dict):
        line["config"]["fixed_blocks_variant"] = _pick(c["fixed_blocks_variant"], "ms_per_step", "content_rays_per_s", "capacity_overflow", "error")
    if isinstance(c.get("prediction"), dict):
        line["config"]["prediction"] = {k: v for k, v in c["prediction"].items() if isinstance(v, (int, float))}
    t = full.get("timing") or {}
    line["timing"] = _pick(t, "ms_per_step_windows", "ms_per_step_median", "ms_per_step_min", "ms_per_step_max")
    if rf:
        line["roofline"] = _pick(rf, "kernel", "bound", "achieved", "peak", "unit", "frac", "frac_l2", "line_utilisation", "frac_hbm_implemented_bytes", "frac_hbm_counters",
                                 "hbm_achieved_GBps", "hbm_peak_GBps", "traffic", "l2_line_bytes", "algorithmic_bytes_per_point", "points_per_launch", "launches", "avg_launch_s",
                                 "counters_round", "counters_commit")
        ws = rf.get("whole_step") or {}
        line["roofline"]["whole_step"] = _pick(ws, "implemented_bytes_per_step", "achieved_GBps", "frac")
    if "roofline_mlp" in full:
        line["roofline_mlp"] = _pick(full["roofline_mlp"], "kernel", "bound", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_s", "mfma_per_32_points")
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {**_pick(cb, "value", "unit", "cores", "kind", "points_per_s", "host_cpus"),
                                "sample": "oracle train step fwd+bwd: 1 warm-up + 3 timed steps of 4096 content rays + 32 blocks at 4x4 of 12x12 rays; same_basis: 1 full step",
                                "same_basis": _pick(cb.get("same_basis") or {}, "value", "unit", "full_step_s", "points", "rays")}
    q = full.get("quality")
    if isinstance(q, dict):
        line["quality"] = _pick(q, "steps", "bit_acc", "wrong_bits_worst_message", "psnr_db", "train_ms_per_step", "bit_acc_before_training", "n_messages", "overflowed", "error")
    sec = full.get("secondary")
    if isinstance(sec, dict):
        out = {}
        for name, v in sec.items():
            if not isinstance(v, dict):
                continue
            if "error" in v:
                out[name] = {"error": str(v["error"])[:80]}
            elif name == "stage1":
                s1 = _pick(v, "ms_per_step", "rays_per_s", "points_per_step", "recaptures", "capacity_overflow", "grid_refresh_every", "host_reads_per_64_steps")
                s1["sparse_grid"] = _pick(v.get("sparse_grid") or {}, "ms_per_step", "points_per_step", "rays_per_s", "ms_per_step_incl_refresh", "ms_per_step_incl_host_refresh")
                s1["refresh"] = {k: x for k, x in ((v.get("sparse_grid") or {}).get("refresh") or {}).items() if isinstance(x, (int, float))}
                for key in ("roofline_scatter", "roofline_wgrad", "roofline_trace"):
                    if isinstance(v.get(key), dict):
                        s1[key] = _pick(v[key], "bound", "achieved", "peak", "unit", "frac", "avg_launch_s", "algorithmic_bytes_per_point", "points_per_launch", "mfma_frac")
                if isinstance(v.get("exchange"), dict):
                    s1["exchange"] = {mode: {k: x for k, x in e.items() if isinstance(x, (int, float, bool))} for mode, e in v["exchange"].items() if isinstance(e, dict)}
                if isinstance(v.get("counter"), dict):
                    s1["counter"] = _pick(v["counter"], "ms_per_step", "rays_per_s", "points_per_step", "cascades", "sparse_grid_ms_per_step")
                out[name] = s1
            elif name == "rank_emulation":
                out[name] = {k: {m: _pick(r, "ms_per_step", "content_rays_per_s_x_ranks_before_xgmi_latency") for m, r in x.items() if isinstance(r, dict)}
                             for k, x in v.items() if isinstance(x, dict)}
            elif name == "distortion_layer":
                out[name] = {k: _pick(x, "ms_per_step", "bit_acc_clean_blocks", "bit_acc_distorted_blocks") for k, x in v.items() if isinstance(x, dict)}
            else:
                out[name] = _pick(v, "ms_per_step", "content_rays_per_s", "points_per_step", "ms_per_image", "images_per_s", "rays_per_s", "bit_acc", "psnr_db", "train_ms_per_step",
                                  "whole_view_device_loop_ms", "staged_4096_device_loop_ms", "identical_images", "ms_per_step_with_the_references_own_train_step_operators")
        line["secondary"] = out
    line["detail"] = DETAIL_FILE
    line = _r(line)
    # the guard: whatever a future field adds, the line stays under the limit -- the optional blocks go first, the contract's fields never
    for drop in ("secondary", "quality", "timing", "roofline_mlp"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        line.setdefault("dropped_for_length", []).append(drop)
    return line


def emit(line, real_stdout):
    """The complete record -> bench_detail.json (beside this file, and under gpurun_out/ when that exists so that it travels back from a GPU box); the compact line,
    alone, on the real stdout."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    json.dump(line, f, indent=1)
            except OSError as e:
                print(f"[bench] could not write {DETAIL_FILE} under {d}: {e}", file=sys.stderr)
    short = json.dumps(compact_line(line))
    assert len(short) <= LINE_LIMIT, len(short)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(short, flush=True)
    os.dup2(2, 1)


# --------------------------------------------------------------------------------------------- the training-step benchmark

def bench_training(args, scene, real_stdout, secondary=None):
    from nerf_signature_amd import _native as nv
    from nerf_signature_amd import blocks, dp, rays, synthetic, trainer
    from nerf_signature_amd.network import NeRFNetwork
    from nerf_signature_amd.optim import CodebookAdam

    if os.environ.get("NERFSIG_DIST_BACKEND", "") != "gloo" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dp.assert_distinct_devices()          # environment only, before the first GPU call: two ranks must never share a device over RCCL
    rank, world, local_rank = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (modulo: a 1-GPU box rehearses N ranks over gloo)
    torch.cuda.set_device(dev)
    cfg = synthetic.SCENES[scene]
    D = cfg["message_dim"]
    H, W = cfg["H"], cfg["W"]
    intr = (cfg["focal"], cfg["focal"], W / 2, H / 2)

    torch.manual_seed(0)
    model = NeRFNetwork(bound=cfg["bound"], cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
    synthetic.init_model(model, scene)
    model.to(dev).train()
    render_kwargs = dict(dt_gamma=cfg["dt_gamma"], max_steps=1024)

    bo, bd = synthetic.block_rays(scene, dev)
    # the poses the per-step content batches come from, and their clean renders ("ground truth" = the clean model's render of the
    # same pose, nerf/provider_wtmk.py:408-416): resident before the timed region
    prng = np.random.RandomState(77)
    poses = torch.from_numpy(np.stack([synthetic.orbit_pose(0.6 + 0.9 * prng.rand(), 2 * np.pi * prng.rand(), cfg["radius"]) for _ in range(args.poses)])).to(dev)
    with torch.no_grad():
        clean = blocks.clean_render(model, poses, intr, H, W, render_kwargs, max_ray_batch=H * W).reshape(args.poses, H * W, 3).clamp_(0, 1).contiguous()
    torch.manual_seed(1000 + rank)          # pixel draws (torch.randint on the device, utils_wtmk_disen.py:105) differ per rank

    def draw_content(step_index):
        """A new pose + new pixels: rays on the device (N1, rg_get_rays) and the matching ground-truth pixels."""
        k = (step_index * world + rank) % args.poses
        r = rays.get_rays(poses[k:k + 1], intr, H, W, N=args.rays)
        return {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "images": clean[k][r["inds"][0]].unsqueeze(0)}

    first = draw_content(0)
    data = {"watermark": {"rays_o_block": bo, "rays_d_block": bd}, "content": first}
    # graph mode: the captured step draws its own batch from the device-resident pose / image store (rg_sample_rays, one launch)
    sampler = None if (args.no_graph or args.fixed_rays or args.host_rays) else rays.DeviceRaySampler(poses, clean, intr, H, W, args.rays, stride=world, offset=rank,
                                                                                                      seed=1000 + rank)
    # main_nerf_wtmk.py:110: Adam(get_params(lr), betas=(0.9, 0.99), eps=1e-15) -- same semantics, the codebook update fused
    optimizer = CodebookAdam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, **({} if args.no_graph else ({"capturable": True} if args.no_fused_adam else {"fused": True, "capturable": True})))
    # README.md:45 (--lambda_w 0.005 --lambda_i 1.0 --iters 1000) and main_nerf_wtmk.py:115 (lr decayed every step): no effect on the timing beyond the
    # per-step learning-rate write, but these are the hyper-parameters the `quality` record below is trained with
    hp = dict(lambda_w=0.005, lambda_i=1.0)
    lr_lambda = lambda it: 0.1 ** min(it / 1000, 1)
    if args.no_graph:
        loop = trainer.WatermarkLoop(model, optimizer, render_kwargs, side_stream=None if args.no_overlap else torch.cuda.Stream(), lr_scheduler=torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda), **hp)
        if args.fixed_blocks:      # the eager loop (what the reference's own Trainer drives): the one-call form of the same declaration
            blk_o, blk_d, _ = trainer.local_blocks(data["watermark"])
            model.fix_rays(blk_o, blk_d, render_kwargs["dt_gamma"], render_kwargs["max_steps"])
    else:
        loop = trainer.GraphedWatermarkLoop(model, optimizer, render_kwargs, data, overlap_content=not args.no_overlap, content_headroom=0.25, content_sampler=sampler,
                                            fixed_blocks=args.fixed_blocks, lr_lambda=lr_lambda, **hp)

    timer = NativeTimer(nv)
    msg_rng = np.random.RandomState(1234)   # same stream on every rank: the message is replicated
    draw = lambda: torch.from_numpy(msg_rng.randint(0, 2, D).astype(np.float32))   # fresh message per step (:1165), host side
    upcoming = [draw()]
    counter = [0]

    running = [loop]        # (the loop one_step drives: the headline loop, later the fixed-blocks variant)

    def one_step():
        # the captured loop is told the next step's message one step early (the same sequence of draws, looked ahead by one):
        # its optimiser kernel then leaves that message's pre-summed codebook behind (GraphedWatermarkLoop, presum_in_adam)
        loop = running[0]
        msg = upcoming.pop()
        upcoming.append(draw())
        counter[0] += 1
        if args.fixed_rays or sampler is not None:
            return loop.step(data, msg) if args.no_graph else loop.step(msg, next_message=upcoming[0])
        if args.no_graph:
            return loop.step({"watermark": data["watermark"], "content": draw_content(counter[0] - 1)}, msg)
        return loop.step(msg, data={"content": draw_content(counter[0] - 1)} if counter[0] > 1 else None, next_message=upcoming[0])

    for _ in range(args.warmup):
        one_step()
    recaptured = False if args.no_graph else bool(loop.ensure_capacity())       # (a host read; outside the timed region)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = args.no_graph          # a replayed graph runs no Python: kernels are timed in the eager pass below
    overflow = False
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    overflow = None if args.no_graph else bool(loop.overflowed())
    # more windows of the same K steps (same loop, same process): the contract's numbers come from the first; the spread says how big a
    # round-to-round difference has to be before it means anything (boxes of the pool differ by ~2 %)
    window_ms = [elapsed / args.steps * 1e3]
    # per-collective times (HIP events around each eagerly executed collective): in the windows BEHIND the first -- the contract's K steps stay untouched -- for the
    # segmented mode, whose collectives run between the graph segments of every replay; collectives captured inside the graph are timed in the eager pass below
    coll_us = {}
    if dp.exchange_active():
        dp.time_collectives(True)
    for _ in range(max(0, args.windows - 1)):
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for i in range(args.steps):
            out = one_step()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        window_ms.append((time.perf_counter() - tw) / args.steps * 1e3)
    if dp.exchange_active():
        coll_us["between_graph_segments_of_the_replays"] = dp.collective_times_us()
        dp.time_collectives(False)
    loss_value = float(out[5].detach())
    loss_parts = (float(out[3].detach()), float(out[4].detach()))
    # Replicated quantities must be the same number on every rank: the watermark loss (every rank decodes the same all-gathered blocks) and
    # the pre-summed codebook (all-reduced, or built from tables every rank updated with the same all-reduced gradient).  A run whose
    # collectives did not do their job ends non-zero (the launcher then starts the next, more conservative attempt).
    ranks_agree = None
    if dist.is_initialized() and world > 1:
        S_now = model._presum_cache[1] if getattr(model, "_presum_cache", None) else None
        probe = torch.stack([out[4].detach().double().reshape(()), (S_now.double().abs().sum() if S_now is not None else torch.zeros((), dtype=torch.float64, device=dev))])
        hi, lo = probe.clone(), probe.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        ranks_agree = bool(((hi - lo).abs() <= 1e-6 * hi.abs() + 1e-12).all().item()) and bool(torch.isfinite(hi).all().item())
        if not ranks_agree:
            print(f"[bench] rank {rank}: replicated quantities differ between the ranks (max {hi.tolist()}, min {lo.tolist()})", file=sys.stderr)

    if not args.no_graph:
        # the same kernels, launched eagerly so that HIP events can bracket them (live, same process, same inputs)
        # ... on ONE stream, back to back: the events then bracket each kernel alone, with the device busy before and after it.  (With the
        # step's streams the content render's kernels run beside the block render's: the MLP forward read 65 us instead of 57.  Draining the
        # device in front of each timed launch instead reads the encoder 10 % SLOWER, 284 us against 258: it starts on an idle device.)
        streams = (loop.side_stream, loop.plan_stream, loop.weights_stream, loop.content_backward_first)
        loop.side_stream = loop.plan_stream = loop.weights_stream = None
        loop.content_backward_first = False
        timer.enabled = True
        if dp.exchange_active():
            dp.time_collectives(True)
        for _ in range(5):
            optimizer.zero_grad(set_to_none=True)
            loop._forward_backward()
            loop.exchange(loop.sink.G)
            loop._optimise_and_march()
        torch.cuda.synchronize()
        timer.enabled = False
        if dp.exchange_active():
            coll_us["eager_pass_one_stream"] = dp.collective_times_us()
            dp.time_collectives(False)
        loop.side_stream, loop.plan_stream, loop.weights_stream, loop.content_backward_first = streams

    if args.no_graph:
        n_block, n_content = int(model.step_counter[(model.local_step - 2) % 16, 0]), int(model.step_counter[(model.local_step - 1) % 16, 0])
        if not args.no_overlap:      # counters are written in issue order: the overlapped step issues the content render first
            n_block, n_content = n_content, n_block
    else:
        n_block, n_content = loop.point_counts()

    # ---- secondary figure, same process, same model: the step with the watermark-block rays declared constant (they are one pair of
    # tensors per dataset, nerf/provider_wtmk.py:442-494): marched once, base-level planes and scatter plan kept, only the codebook level
    # gathered per step (GraphedWatermarkLoop(fixed_blocks=True); bit-identical renders, tests/test_gpu_fixed.py).  NOT the headline:
    # `value` above is the step that recomputes everything every step, like the reference.
    variant = None
    if world == 1 and scene == "hotdog" and not (args.no_graph or args.fixed_blocks or args.fixed_rays or args.host_rays or args.no_variant) and not dp.exchange_active():
        try:
            loop.close()
            loop2 = trainer.GraphedWatermarkLoop(model, optimizer, render_kwargs, data, overlap_content=not args.no_overlap, content_headroom=0.25, content_sampler=sampler,
                                                 fixed_blocks=True, lr_lambda=lr_lambda, **hp)
            running[0] = loop2
            for _ in range(max(args.warmup, 2)):
                one_step()
            loop2.ensure_capacity()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                out2 = one_step()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            variant = {"what": "watermark-block rays declared constant (one pair of tensors per dataset, provider_wtmk.py:442-494): samples marched once, their 16 base-level "
                               "feature planes and scatter plan kept, per step only the codebook level is gathered for them; MLPs, compositing, decoder, backward, scatter and "
                               "optimiser run every step; renders bit-identical to the recomputing step (tests/test_gpu_fixed.py); opt-in: GraphedWatermarkLoop(fixed_blocks=True), "
                               "bench.py --fixed-blocks",
                       "ms_per_step": el2 / args.steps * 1e3, "content_rays_per_s": args.rays * args.steps / el2, "steps": args.steps,
                       "loss": float(out2[5].detach()), "capacity_overflow": bool(loop2.overflowed())}
        except Exception as e:      # a secondary figure must not take the headline down
            variant = {"error": repr(e)}
    sharded = bool(getattr(loop, "sharded", False)) or (args.no_graph and dp.block_shard(bo.shape[0]) is not None)
    rays_block_all = bo.shape[0] * bo.shape[1] * bo.shape[2]
    rays_block_rank = rays_block_all // world if sharded else rays_block_all
    rays_content = args.rays
    # the dominant kernel: the hash-gather encoder, on the launch with the most points (N = 1: the block render)
    big = max(n_block, n_content) // 2
    enc_big = big
    enc_s, enc_n, enc_rows = timer.stats("hg_encode_planes", enc_big)
    if enc_n == 0:      # --fixed-blocks: the block render no longer runs the 17-level encoder per step; the content render's launch is the largest left
        enc_big = 0
        enc_s, enc_n, enc_rows = timer.stats("hg_encode_planes", enc_big)
    mlp_s, _, mlp_rows = timer.stats("field_fwd", big)
    bwd_s, _, _ = timer.stats("field_bwd_planned", big)          # the block render goes through the planned binned route
    sct_s, _, _ = timer.stats("hg_scatter_planned", big)
    plan_s, _, _ = timer.stats("hg_scatter_plan", big)           # (beside the forward pass: not on the critical path)
    if bwd_s == 0.0:
        bwd_s, _, _ = timer.stats("field_bwd", big)
        sct_s, _, _ = timer.stats("hg_scatter_binned", big)
        if sct_s == 0.0:
            sct_s, _, _ = timer.stats("hg_scatter_sliced", big)
    pts_big = float(max(n_block, n_content)) if enc_big else float(n_content)
    pts_step = float(n_block + n_content)
    ms = elapsed / args.steps * 1e3

    if rank != 0:
        if ranks_agree is False:
            raise SystemExit(4)
        return
    # ---- bytes.  IMPLEMENTED algorithm: the D selected codebook tables are pre-summed into one (linearity of the
    # trilinear interpolation), so a point gathers 16 base levels + 1 summed level, 8 corners x 8 B each.
    gather_impl = 16 * 64 + 64                     # 1088 B/point actually gathered by k_encode_planes
    split_encoder_early = bool(getattr(loop, "fixed_blocks", False))
    split_encoder = split_encoder_early and enc_big != 0
    gather_launch = 16 * 64 if split_encoder else gather_impl   # (--fixed-blocks: the timed launch gathers the 16 base levels; the codebook level is its own 64 B/point launch)
    gather_ref = 16 * 64 + 64 * D                  # the reference algorithm's D separate codebook gathers (side value only)
    achieved = pts_big * gather_launch / enc_s if enc_s > 0 else 0.0
    traffic, l1_model, traffic_source = None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:   # HBM bytes per launch of the same kernel on the same inputs, from a separate rocprofv3 --pmc run (profiles/)
            counters = json.load(open(pmc))
            traffic = counters.get("k_encode_planes_hbm_bytes_per_launch")
            # NOT measured by this run: counters need their own rocprofv3 passes (tools/pmc_step.sh); the file says at which round / commit they were taken
            traffic_source = {"file": "profiles/pmc_traffic.json", "round": counters.get("round"), "commit": counters.get("commit"),
                              "context": "the launch at the head of the step, behind the optimiser + warm-up pass (tools/pmc_step.py)" if counters.get("round") else "stand-alone launches (rounds 1-2)"}
            # what actually bounds the launch: the CU's vector-L1 pipe looks up one line per clock and fills a missing one
            # in 2.4 clk (tools/micro/gather_rate.hip); counters of the block render's launch (17 levels, 1.29 M points)
            enc = dict(counters.get("k_encode_planes", {}))
            # the counters were taken on a launch of `k_encode_planes_points_per_launch` points (the one-GPU block render: 1.29 M); a rank of R renders 1/R of the blocks --
            # lookups, fills and HBM bytes of the gather scale with the points
            at_points = counters.get("k_encode_planes_points_per_launch")
            if at_points and enc_big and abs(pts_big - at_points) > 0.01 * at_points:
                k_pts = pts_big / at_points
                for name in ("TCP_TOTAL_CACHE_ACCESSES", "TCP_TCC_READ_REQ"):
                    if enc.get(name):
                        enc[name] = enc[name] * k_pts
                traffic = traffic * k_pts if traffic else traffic
                traffic_source["scaled_to_this_launch"] = k_pts
            if enc.get("TCP_TOTAL_CACHE_ACCESSES") and enc.get("TCP_TCC_READ_REQ") and enc_big and not split_encoder_early:
                clk = (enc["TCP_TOTAL_CACHE_ACCESSES"] + 1.4 * enc["TCP_TCC_READ_REQ"]) / 256.0
                l1_model = {"tag_lookups_per_launch": enc["TCP_TOTAL_CACHE_ACCESSES"], "misses_to_L2_per_launch": enc["TCP_TCC_READ_REQ"],
                            "model_s_at_2.4GHz": clk / 2.4e9, "model": "(lookups x 1 clk + misses x 1.4 clk) / 256 CUs",
                            "measured_over_model": (enc_s / (clk / 2.4e9)) if enc_s > 0 else None}
        except Exception:
            traffic = None
    binding_roof = {"l2_line_bytes": None, "frac_l2": None, "line_utilisation": None}
    if l1_model is not None and enc_s > 0:
        fills = l1_model["misses_to_L2_per_launch"]
        binding_roof = {"l2_line_bytes": fills * 128.0, "l2_peak_GBps": L2_PEAK / 1e9, "l2_achieved_GBps": fills * 128.0 / enc_s / 1e9, "frac_l2": fills * 128.0 / enc_s / L2_PEAK,
                        "line_utilisation": pts_big * gather_launch / (fills * 128.0),
                        "binding_roof_note": "fills x 128 B = bytes moved L2 -> L1 for `algorithmic_bytes_per_point` x points used: the reference's hash sends the 8 corners of a point "
                                             "to ~4 different lines on every hashed level (profiles/r02_encoder_experiments.txt), so ~1/4 of every line is used; frac_l2 is the "
                                             "fraction of the aggregate L2 -> L1 rate, the roof that binds this kernel (not HBM: `frac` above is an accounting figure)"}
    # whole step, implemented bytes: per point forward 1088 gathered + 32 (xyz, dir, deltas) + 16 (sigma, rgb); backward 64 (upstream
    # gradients, saved sigma/rgb/masks) + 128 (four 16-byte scatter-queue entries written and read); per step the optimiser's streams:
    # G once, param/exp_avg/exp_avg_sq of D tables read+written (6 D tables), ~D/2 partner tables + S written for the next pre-sum
    step_bytes = pts_step * (gather_impl + 48 + 64 + 128) + (1 + 6 * D + D / 2 + 1) * T_BYTES
    line = {
        "metric": "training rays/sec @4096 rays (hotdog, 32-bit msg)" if scene == "hotdog" else f"training rays/sec @4096 rays ({scene}, {D}-bit msg)",
        "value": rays_content * world * args.steps / elapsed,
        "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms,
        "timing": {"windows": len(window_ms), "steps_per_window": args.steps, "ms_per_step_windows": [round(w, 5) for w in window_ms], "ms_per_step_median": float(np.median(window_ms)),
                   "ms_per_step_min": float(np.min(window_ms)), "ms_per_step_max": float(np.max(window_ms)),
                   "note": "`value` and `ms_per_step` are the FIRST window (exactly --steps steps between barrier + synchronize); the others repeat it in the same process (this rank's clock)"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (hash gather, compositing) + " + nv.mlp_precision_name(),
        "data": "synthetic",
        "config": {
            "workload": f"Blender/hotdog-like synthetic scene S0, --wtmk_tcnn, per rank and step: {rays_content} content rays of a new pose + its share of the 32x12x12 "
                        f"watermark-block rays, 32-bit msg, {world}xMI355X HIP raymarch+hash+MLP" if scene == "hotdog" else
                        f"Mip-NeRF360/counter-like synthetic scene S1 (bound 2, two cascades, camera inside), {rays_content} content rays + 32x12x12 block rays, {D}-bit msg, {world}xMI355X",
            "value_basis": "content rays (the metric's 4096-ray batch) x ranks / time; block rays excluded",
            "all_rays_per_s": (rays_content * world + rays_block_all) * args.steps / elapsed,
            "points_per_s": (n_content * world + (n_block * world if sharded else n_block)) * args.steps / elapsed,
            "content_rays": rays_content, "block_rays_total": rays_block_all, "block_rays_this_rank": rays_block_rank, "blocks_sharded_over_ranks": sharded,
            "points_per_step_per_rank": n_block + n_content, "samples_per_ray_block": n_block / max(rays_block_rank, 1),
            "samples_per_ray_content": n_content / rays_content,
            "ray_sets": "fixed" if args.fixed_rays else f"new pose (of {args.poses} pre-rendered clean views) + new random pixels every step, " +
                        ("drawn inside the captured step from the device-resident pose/image store (rg_sample_rays)" if sampler is not None else "rays generated on the device inside the timed loop (host-driven: randint, rg_get_rays, gather, copies)"),
            "content_march": "head of the step",
            "message_dim": D, "parallelism": f"dp{world}", "optimizer": "Adam(betas=(0.9,0.99), eps=1e-15): torch semantics, codebook update fused (opt_codebook_adam)",
            "hyper_parameters": "README.md:45: lambda_w 0.005, lambda_i 1.0, lr 1e-2 * 0.1 ** min(it / 1000, 1) written every step",
            "grad_exchange_bytes_per_step": loop.exchange.bytes_per_step + (rays_block_all * 12 if sharded else 0) + (T_BYTES if getattr(loop, "opt_shard", None) else 0),
            "collectives_per_step": (loop.exchange.collectives_per_step + (1 if sharded else 0) + (1 if getattr(loop, "opt_shard", None) else 0)) if dp.exchange_active() else 0,
            "codebook_optimizer": ("sharded over the ranks (each updates the tables of D/R bits, partial pre-sums all-reduced)" if getattr(loop, "opt_shard", None) else "replicated"),
            "world_size_seen_by_backend": dp.world_size(), "backend": dist.get_backend() if dist.is_initialized() else None,
            "launch_attempt": args.launch_attempt or "n/a (single process or external launcher)",
            "collective_us": coll_us or None,
            "collective_us_note": "HIP events on the issuing stream around every eagerly executed collective, rank 0: `between_graph_segments_of_the_replays` in the windows behind the first "
                                  "(segmented mode only: captured collectives are graph nodes), `eager_pass_one_stream` in the 5 eager steps behind the timed run; the times include the wait for "
                                  "the slowest rank" if coll_us else None,
            "prediction": prediction(world, "captured" if (not args.no_graph and len(loop.segments) == 1) else "segmented", ms) if world > 1 else None,
            "watchdog_drain_before_capture": dp.LAST_DRAIN,
            "ranks_agree_on_replicated_values": ranks_agree,
            "execution": "eager" if args.no_graph else f"hipGraph replay, {len(loop.segments)} captured segment(s)" + (" with the RCCL collectives between them" if len(loop.segments) > 1 else
                                                                  (" with the RCCL collectives captured inside" if dp.exchange_active() else "")),
            "capacity_overflow": overflow, "recaptured_with_more_headroom_after_warmup": recaptured,
            "block_rays_policy": "declared constant: marched once, base-level planes + scatter plan kept (--fixed-blocks)" if getattr(loop, "fixed_blocks", False) else
                                 "recomputed every step (march, 17-level gather, scatter plan), like the reference",
            "loss": loss_value, "loss_image": loss_parts[0], "loss_watermark": loss_parts[1],
        },
        "roofline": {
            "kernel": "k_encode_planes",
            # what the counters show bounds this launch: the CU's vector-L1 (tag lookups + L2->L1 line fills), not HBM -- the 68 MiB working set is
            # L2 / Infinity-Cache resident.  `achieved` / `peak` / `frac` stay priced against the HBM peak (the figure north_star's ">= 40 % HBM-read
            # roofline on the hash gather" names): frac == frac_hbm_implemented_bytes; the HBM bytes the launch really moves are frac_hbm_counters.
            # THE BINDING ROOF LEADS: with counters on file, `bound` = "l2_l1_fill" and achieved / peak / frac are the L2 -> L1 line-fill rate
            # against the aggregate L2 peak; the HBM accounting figure north_star names (implemented bytes / time / 8 TB/s) rides along as
            # hbm_achieved_GBps / hbm_peak_GBps / frac_hbm_implemented_bytes.  Without counters the record falls back to the HBM pricing.
            "bound": "l2_l1_fill" if binding_roof.get("frac_l2") else "hbm",
            "achieved": binding_roof["l2_achieved_GBps"] if binding_roof.get("frac_l2") else achieved / 1e9,
            "peak": L2_PEAK / 1e9 if binding_roof.get("frac_l2") else HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": binding_roof["frac_l2"] if binding_roof.get("frac_l2") else achieved / HBM_PEAK,
            "hbm_achieved_GBps": achieved / 1e9, "hbm_peak_GBps": HBM_PEAK / 1e9,
            "frac_hbm_implemented_bytes": achieved / HBM_PEAK,
            "counters_round": (traffic_source or {}).get("round"), "counters_commit": (traffic_source or {}).get("commit"),
            "frac_of_measured_copy_ceiling": achieved / 6.29e12,      # (6.29 TB/s: the stream-copy rate MI355X_MICROARCH.md measures)
            "traffic": traffic, "traffic_source": traffic_source,
            # the SURVEY 8(d) formula's bytes (D separate codebook gathers: 3072 B/point) over the same time: > 1 because the kernel does not move them
            # (pre-summed codebook) -- both bases side by side in the driver's record
            "reference_algorithm_frac": (pts_big * gather_ref / enc_s / HBM_PEAK) if (enc_s > 0 and not split_encoder) else None,
            "launches": enc_n, "avg_launch_s": enc_s, "points_per_launch": pts_big, "rows_per_launch": enc_rows,
            "algorithmic_bytes_per_point": gather_launch,
            "basis": "ACCOUNTING vs HBM -- the kernel is L1/L2-bound, see frac_l2 / line_utilisation; " + ("bytes of the IMPLEMENTED algorithm, this launch: the 16 base levels, 8 corners x 8 B (the pre-summed codebook level is gathered by its own launch at the "
                      "head of the step, k_encode_codebook_plane: 64 B/point, timed below)" if split_encoder else
                      "bytes of the IMPLEMENTED algorithm: 16 base levels + the pre-summed codebook level, 8 corners x 8 B (DESIGN.md sections 2 and 6)"),
            "codebook_level_launch_s": timer.stats("hg_encode_codebook_plane", big)[0] if split_encoder else None,
            "scheduling": "head of the step; timed in 5 eager steps issued on one stream (no kernel beside it)",
            "observed_limiter": "not HBM: the working set (64 MiB base + 4 MiB pre-sum) is L2/MALL-resident; PMC shows the texture-address path busy ~85 % and "
                                "the L2->L1 line fills (~4 GB per launch) as the limiter (profiles/*pmc_encode*)",
            "frac_hbm_counters": (traffic / enc_s / HBM_PEAK) if (traffic and enc_s > 0) else None,
            # THE BINDING ROOF: the launch is bound by L2 -> L1 line fills.  Fills from the counters (TCP_TCC_READ_REQ, one 128-byte line
            # each) over THIS run's launch time, against the aggregate L2 rate the guide measures (34.5 TB/s, MI355X_MICROARCH.md "L2 (per XCD)").
            **binding_roof,
            "l1_lookup_rate_model": l1_model,
            "reference_algorithm": {"bytes_per_point": gather_ref, "note": "SURVEY.md 8(d) formula (D separate codebook gathers); the kernel does not move these bytes, "
                                    "so this is a speed-up factor over a literal implementation, NOT a roofline fraction",
                                    "equivalent_GBps": (pts_big * gather_ref / enc_s / 1e9) if enc_s > 0 else 0.0},
            "forward_encoder_plus_mlp_s": enc_s + mlp_s,
            "backward_mlp_plus_scatter": {"avg_s": bwd_s + sct_s, "k_field_bwd_s": bwd_s, "scatter_s": sct_s, "plan_s_off_path": plan_s},
            "whole_step": {"implemented_bytes_per_step": step_bytes, "achieved_GBps": step_bytes / (ms * 1e-3) / 1e9, "frac": step_bytes / (ms * 1e-3) / HBM_PEAK,
                           "note": "all kernels of the step (march, decoder, compositing, losses included in the time, not in the bytes)"},
        },
    }
    # the MLP kernel against the matrix-core roof (north_star: "MFMA utilisation on the MLP against chip peak").
    rows = float(mlp_rows) if mlp_rows else pts_big
    mfma_per_wave, kind = nv.mlp_mfma_per_wave()
    issued_flop = rows / 32.0 * mfma_per_wave * 32768
    line["roofline_mlp"] = {
        "kernel": f"k_field_fwd<planes> (sigma MLP + SH + colour MLP, {nv.mlp_precision_name()}) on the same launch",
        "bound": "mfma", "achieved": (issued_flop / mlp_s / 1e12) if mlp_s > 0 else 0.0, "peak": MFMA_PEAK[kind] / 1e12, "unit": "TFLOP/s",
        "frac": (issued_flop / mlp_s / MFMA_PEAK[kind]) if mlp_s > 0 else 0.0, "avg_launch_s": mlp_s,
        "algorithmic_TFLOPs": (float(max(n_block, n_content)) * 20480 / mlp_s / 1e12) if mlp_s > 0 else 0.0,
        "frac_algorithmic": (float(max(n_block, n_content)) * 20480 / mlp_s / MFMA_PEAK[kind]) if mlp_s > 0 else 0.0,
        "mfma_per_32_points": mfma_per_wave,
        "note": "achieved = issued MFMA FLOP/s (v_mfma_f32_32x32x16: 32768 FLOP each); algorithmic = 20 480 FLOP per point (SURVEY.md 8(d))",
    }
    if variant is not None:
        line["config"]["fixed_blocks_variant"] = variant
    if world == 1 and not args.no_cpu_baseline and scene == "hotdog":
        line["cpu_baseline"] = cpu_baseline(model, D, full_step_points=pts_step, threads=args.cpu_threads)
    if secondary is not None:
        quality = secondary.pop("quality", None)
        if quality is not None:      # the reference's whole run on this path (1000 steps, README hyper-parameters) + test_bitacc / test_image: nerf_signature_amd/quality.py
            line["quality"] = quality
        line["secondary"] = secondary
    if ranks_agree is False:
        raise SystemExit(4)
    emit(line, real_stdout)


# --------------------------------------------------------------------------------------------- config 5: full-image eval + decoder

def bench_fern(args, real_stdout):
    """BASELINE.json config 5 (LLFF/fern-like scene S2): 48-bit message, 64x64 block grid, bound 2 / two cascades / dt_gamma 1/128;
    one step = the watermarked full-image render of a 1008x756 view, staged in 4096-ray chunks (187 chunks, renderer_wtmk.py:555-570)
    under no_grad with the model in training mode (utils_wtmk_disen.py:832: test_image never calls model.eval()), then the 48 selected
    11x15 blocks through the HiDDeN decoder and the bit accuracy (eval_step, :648-702)."""
    from nerf_signature_amd import rays, synthetic, trainer
    from nerf_signature_amd.network import NeRFNetwork
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    scene = "fern"
    cfg = synthetic.SCENES[scene]
    D, H, W = cfg["message_dim"], cfg["H"], cfg["W"]
    intr = (cfg["focal"], cfg["focal"], W / 2, H / 2)
    torch.manual_seed(0)
    model = NeRFNetwork(bound=cfg["bound"], cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
    synthetic.init_model(model, scene)
    model.to(dev).train()
    kw = dict(dt_gamma=cfg["dt_gamma"], max_steps=1024)
    pose = torch.from_numpy(synthetic.orbit_pose(1.1, 0.7, cfg["radius"]))[None].to(dev)
    bh, bw = H // cfg["rows"], W // cfg["cols"]
    bo, _ = synthetic.block_rays(scene, dev)
    assert bo.shape[:3] == (D, bh, bw)
    msg = torch.from_numpy(np.random.RandomState(5).randint(0, 2, D).astype(np.float32))
    acc = trainer.BIT_ACC()
    r = rays.get_rays(pose, intr, H, W, -1)
    bo_all, bd_all = synthetic.block_rays(scene, dev)

    @torch.no_grad()
    def one_image():
        img = model.render(r["rays_o"], r["rays_d"], msg, staged=True, max_ray_batch=4096, bg_color=1, perturb=False, force_all_rays=True, **kw)["image"]
        blocks_img = model.render(bo_all, bd_all, msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **kw)["image"]
        decoded = model.msg_decoder(model.normalization(blocks_img.clamp(0, 1).permute(0, 3, 1, 2)))
        return img, decoded

    for _ in range(max(args.warmup, 1)):
        one_image()
    torch.cuda.synchronize()
    steps = max(1, min(args.steps, 10))
    t0 = time.perf_counter()
    for _ in range(steps):
        img, decoded = one_image()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    acc.update(decoded.permute(1, 0), msg[None].to(dev))
    n_chunks = (H * W + 4095) // 4096
    emit({"metric": "full-image watermarked render + decoder eval (fern, 48-bit msg)", "value": steps / elapsed, "unit": "images/s", "n_gpus": 1, "steps": steps,
          "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
          "dtype": "f32 (hash gather, compositing) + " + __import__("nerf_signature_amd._native", fromlist=["x"]).mlp_precision_name(), "data": "synthetic",
          "config": {"workload": f"LLFF/fern-like synthetic scene S2: {W}x{H} view staged in {n_chunks} chunks of 4096 rays (training-mode kernels under no_grad, "
                                 f"as the reference's test_image runs them) + {D} blocks of {bh}x{bw} through the HiDDeN decoder, 48-bit msg, 64x64 block grid, 1xMI355X",
                     "rays_per_s": (H * W + D * bh * bw) * steps / elapsed, "chunks": n_chunks, "bit_accuracy_random_init": acc.measure(),
                     "image_mean": float(img.mean()), "secondary": True}}, real_stdout)


def main():
    args = parse_args()
    in_rank = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_rank:
        launch_ranks(args)                       # never returns
    if (args.gpus > 1 and in_rank and not args.launched_by and os.environ.get("NERFSIG_CAPTURE_COLLECTIVES") not in ("0", "1")
            and not args.no_graph and int(os.environ.get("WORLD_SIZE", "1")) > 1):
        supervise_rank(args)                     # an external launcher started this rank: never returns
    if args.dry_launch:
        dry_launch(args)                         # never returns
    # stdout carries exactly one JSON line: RCCL prints its version banner to stdout when the process group starts, so
    # everything until the result goes to stderr's descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    secondary = None
    if (args.gpus == 1 and args.config == "hotdog" and not in_rank and not (args.no_secondary or args.no_graph or args.fixed_blocks or args.fixed_rays or args.host_rays)):
        secondary = run_secondaries(args)        # child processes, BEFORE anything here touches the GPU
    if args.config == "fern":
        bench_fern(args, real_stdout)
    else:
        bench_training(args, args.config, real_stdout, secondary)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
