#!/usr/bin/env python
"""Stage-1 (clean model) training step on the bench scene: 4096 rays, all parameters trainable.

    python tools/stage1_bench.py [content|block] [--eager] [--steps K] [--windows W] [--no-refresh] [--no-overlap | --overlap] [--two-launch] [--three-launch-composite] [--separate-table-adam] [--json]

Default: the captured loop (stage1.GraphedCleanLoop), perturbed samples, the density grid refreshed every 16 steps INSIDE the timed
windows (the reference's loop does it there, nerf/utils.py:852-857).  W windows of K steps each; a window in which the loop had to grow
its buffers and capture again is reported but left out of the median.  Behind the windows the same explicit kernel sequence is issued
eagerly on ONE stream with HIP events around every entry point (nothing runs beside the timed kernel): the per-kernel table and the two
roofline records -- the 16-level table scatter and the weight-gradient reduction.
--json: one JSON line on stdout (bench.py's `secondary.stage1`).
--two-launch: field_bwd_trace + field_wgrad (the first version's backward) instead of the one-launch field_bwd_wgrad.
--eager: the autograd loop (stage1.CleanLoop) with the per-entry-point breakdown."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from nerf_signature_amd import _native as nv
from nerf_signature_amd import synthetic
from nerf_signature_amd.stage1 import CleanLoop, CleanNeRFNetwork, GraphedCleanLoop

HBM_PEAK, ATOMIC_PEAK, MFMA_PEAK_BF16 = 8.0e12, 1.3e12, 2.5e15      # B/s spec; B/s of added bytes (MI355X_MICROARCH.md "Global float atomics"); dense FLOP/s


def flag(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit() and a not in synthetic.SCENES]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
which = args[0] if args else "content"
steps, windows = flag("--steps", 64), flag("--windows", 5)
as_json = "--json" in flags
if "--rccl1" in flags:      # a world-size-1 RCCL group: the step's all-reduce of [16 table gradients | MLP gradients] (64 MiB) issued for real -- between two captured
    # segments, or (NERFSIG_CAPTURE_COLLECTIVES=1) inside the one graph
    os.environ.update(NERFSIG_FORCE_EXCHANGE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    from nerf_signature_amd import dp
    dp.init_from_env()
say = (lambda *a: print(*a, file=sys.stderr)) if as_json else print
dev = torch.device("cuda")


scene = sys.argv[sys.argv.index("--scene") + 1] if "--scene" in sys.argv else "hotdog"      # hotdog = S0 (bound 1); counter = S1 (bound 2, two cascades, camera inside: BASELINE config 3)
BOUND = synthetic.SCENES[scene]["bound"]


def fresh_model():
    m = CleanNeRFNetwork(bound=BOUND, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
    with torch.no_grad():
        for l, e in enumerate(m.encoder.embeddings):
            e.weight.copy_(torch.from_numpy(synthetic.table_values(l, 0.5)))
        grid = synthetic.density_grid(BOUND)
        bits, _ = synthetic.pack_bits_np(grid, 10.0)
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bits))
    return m.to(dev).train()


torch.manual_seed(0)
m = fresh_model()
if which == "block":
    o, d = synthetic.block_rays(scene, dev)
    o, d = o.reshape(1, -1, 3), d.reshape(1, -1, 3)
else:
    o, d = synthetic.content_rays(scene, 4096, 0, dev)
target = torch.rand(1, o.shape[1], 3, device=dev)
n_rays = o.shape[1]
opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, **({"fused": True} if "--eager" in flags else {}))
data = {"rays_o": o, "rays_d": d, "images": target, "perturb": False, "force_all_rays": True}


class CallTimer:
    """HIP events around every libnerfsig entry point (on the stream the kernels are launched on)."""

    def __init__(self):
        self.events, self.orig = {}, nv.call

    def __call__(self, name, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.orig(name, *a)
        e1.record()
        self.events.setdefault(name, []).append((e0, e1))

    def __enter__(self):
        nv.call = self
        return self

    def __exit__(self, *exc):
        nv.call = self.orig

    def us(self, name, per):
        return sum(a.elapsed_time(b) for a, b in self.events.get(name, [])) / per * 1e3

    def table(self, per):
        return sorted(((k, len(v) / per, self.us(k, per)) for k, v in self.events.items()), key=lambda r: -r[2])


if "--eager" in flags:
    loop = CleanLoop(m, opt, dict(dt_gamma=0, max_steps=1024), update_extra_interval=10 ** 9)
    loop.global_step = 1
    for _ in range(3):
        loop.step(data)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    with CallTimer() as timer:
        t0.record()
        for _ in range(n):
            loop.step(data)
        t1.record()
        torch.cuda.synchronize()
    pts = int(m.step_counter[(m.local_step - 1) % 16, 0])
    ms = t0.elapsed_time(t1) / n
    print(f"stage-1 eager step ({which}): {o.shape[1]} rays, {pts} points: {ms:.3f} ms/step = {o.shape[1] / ms * 1e3:.3e} rays/s")
    for k, per, us in timer.table(n):
        print(f"  {k:32s} {per:5.1f} launches/step {us:9.1f} us/step")
    sys.exit(0)

refresh = 0 if "--no-refresh" in flags else 16
fused = "--two-launch" not in flags
plan_mode = False if "--no-overlap" in flags else (True if "--overlap" in flags else "auto")      # the scatter plan on a stream of its own: never | always | from 600 k buffer rows on
one_composite = "--three-launch-composite" not in flags      # (rm_composite_train_mse | compositing forward, clean_loss, compositing backward)
table_adam_flag = False if "--separate-table-adam" in flags else None      # (None: the loop's default -- inside the scatter's owners unless gradients are exchanged)
trace_dtype = "f16" if "--f16-traces" in flags else "f32"      # the saved layer inputs as fp16 (half the trace bytes) | fp32 (strict)
host_refresh = "--host-refresh" in flags      # the reference's form of the grid refresh (torch operators, three host synchronisations) instead of the device-side graph
loop = GraphedCleanLoop(m, opt, dict(dt_gamma=0, max_steps=1024), n_rays=n_rays, update_extra_interval=refresh, perturb=True, overlap_plan=plan_mode,
                        fused_backward=fused, fused_composite=one_composite, fused_table_adam=table_adam_flag, device_refresh=not host_refresh, trace_dtype=trace_dtype)
loop.step(data)
for _ in range(31):
    loop.step()
torch.cuda.synchronize()
win = []
for w in range(windows):
    rec0 = loop.recaptures
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        loop.step()
    t1.record()
    torch.cuda.synchronize()
    win.append({"ms_per_step": t0.elapsed_time(t1) / steps, "recaptured": loop.recaptures > rec0, "points_last_step": int(loop.count_ring[(loop.global_step - 1) % 16, 0])})
clean = [w["ms_per_step"] for w in win if not w["recaptured"]] or [w["ms_per_step"] for w in win]
ms = float(np.median(clean))
pts = win[-1]["points_last_step"]
loss_last = loop.losses(1)[0]
overflow = loop.overflowed()
say(f"stage-1 captured step ({which}): {n_rays} rays, {pts} points, capacity {loop.capacity}: median {ms:.3f} ms/step = {n_rays / ms * 1e3:.3e} rays/s over "
    f"{len(clean)} of {windows} windows x {steps} steps ({', '.join(('%.3f' % w['ms_per_step']) + ('*' if w['recaptured'] else '') for w in win)}; * = re-captured inside); "
    f"grid refresh every {refresh} steps inside the windows; loss {loss_last:.4e}; recaptures {loop.recaptures}")

# ---- the same captured loop on the scene's own (sparse) occupancy grid, never refreshed: the sample count a TRAINED scene has (the ball: ~34 samples per ray) instead of
# the full grid a random field leaves behind
sparse = None
if refresh:
    try:
        m_s = fresh_model()
        opt_s = torch.optim.Adam(m_s.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        loop_s = GraphedCleanLoop(m_s, opt_s, dict(dt_gamma=0, max_steps=1024), n_rays=n_rays, update_extra_interval=0, perturb=True, overlap_plan=plan_mode,
                                  fused_backward=fused, fused_composite=one_composite, fused_table_adam=table_adam_flag, trace_dtype=trace_dtype)
        loop_s.step(data)
        for _ in range(15):
            loop_s.step()
        torch.cuda.synchronize()
        ws = []
        for _ in range(3):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(steps):
                loop_s.step()
            t1.record()
            torch.cuda.synchronize()
            ws.append(t0.elapsed_time(t1) / steps)
        pts_s = int(loop_s.count_ring[(loop_s.global_step - 1) % 16, 0])
        sparse = {"ms_per_step": float(np.median(ws)), "windows_ms": ws, "points_per_step": pts_s, "rays_per_s": n_rays / float(np.median(ws)) * 1e3,
                  "what": "the same captured step on the scene's own occupancy grid, no refresh (the sample count of a trained scene: the ball, ~34 samples per ray)"}
        say(f"  on the sparse grid (no refresh): {pts_s} points, median {sparse['ms_per_step']:.3f} ms/step = {sparse['rays_per_s']:.3e} rays/s")
        # the grid refresh by itself, in both of its forms and both implementations (wall time of one refresh between two device synchronisations; behind the
        # windows: a random field's densities would fill the sparse grid).  Device form: refreshes 0 / 16 run eagerly, 1 / 17 are captured, the rest replay.
        import time

        def time_refreshes(n):
            out = []
            for _ in range(n):
                torch.cuda.synchronize()
                t = time.perf_counter()
                loop_s.refresh_grid()
                torch.cuda.synchronize()
                out.append((time.perf_counter() - t) * 1e3)
            return out

        m_s.iter_density = 0
        loop_s.device_refresh = True
        t_dev = time_refreshes(28)
        m_s.iter_density = 0
        loop_s.device_refresh = False
        t_host = time_refreshes(28)
        sparse["refresh"] = {"device_full_ms": float(np.median(t_dev[2:16])), "device_partial_ms": float(np.median(t_dev[18:])),
                             "host_full_ms": float(np.median(t_host[2:16])), "host_partial_ms": float(np.median(t_host[18:])),
                             "host_reads_per_refresh": {"device": 0, "host": 3}, "every_steps": 16}
        sparse["ms_per_step_incl_refresh"] = sparse["ms_per_step"] + sparse["refresh"]["device_partial_ms"] / 16
        sparse["ms_per_step_incl_host_refresh"] = sparse["ms_per_step"] + sparse["refresh"]["host_partial_ms"] / 16
        say(f"  one grid refresh: device graph {sparse['refresh']['device_full_ms']:.3f} ms (full) / {sparse['refresh']['device_partial_ms']:.3f} ms (partial), "
            f"torch operators {sparse['refresh']['host_full_ms']:.3f} / {sparse['refresh']['host_partial_ms']:.3f} ms -> sparse-grid step incl. refresh every 16 steps: "
            f"{sparse['ms_per_step_incl_refresh']:.3f} ms (was {sparse['ms_per_step_incl_host_refresh']:.3f})")
        loop_s.close()
        del loop_s, m_s, opt_s
    except Exception as e:      # noqa: BLE001 -- a side figure
        sparse = {"error": repr(e)}

# ---- the same kernel sequence, eagerly, on one stream, every entry point between HIP events
n_segments = len(loop.graph.segments) if loop.graph is not None else None
loop_table_adam = bool(loop.fused_table_adam)
steps_done, capacity, state = loop.global_step, loop.capacity, {k: v.detach().clone() for k, v in m.state_dict().items()}
loop.close()
def eager_pass(table_adam_in_owners):
    m2 = fresh_model()
    m2.load_state_dict(state)
    opt2 = torch.optim.Adam(m2.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    eager = GraphedCleanLoop(m2, opt2, dict(dt_gamma=0, max_steps=1024), n_rays=n_rays, update_extra_interval=0, perturb=True, overlap_plan=False, capture=False, capacity=capacity,
                             fused_backward=fused, fused_composite=one_composite, fused_table_adam=table_adam_in_owners, trace_dtype=trace_dtype)
    eager.step(data)
    for _ in range(3):
        eager.step()
    torch.cuda.synchronize()
    with CallTimer() as tm:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            eager.step()
        e1.record()
        torch.cuda.synchronize()
    pts_here = float(eager.count_ring[:, 0].float()[eager.count_ring[:, 0] > 0].mean())
    eager.close()
    return tm, e0.elapsed_time(e1) / n, pts_here


n = 10
table_adam = loop_table_adam
timer, eager_ms, pts_e = eager_pass(table_adam)          # what the captured step runs
rows = timer.table(n)
say(f"  the same sequence issued eagerly on one stream: {eager_ms:.3f} ms/step, {pts_e:.0f} points/step")
for k, per, us in rows:
    say(f"  {k:32s} {per:5.1f} launches/step {us:9.1f} us/step")
fused_scatter_s = timer.us("hg_levels_scatter_adam", n) * 1e-6 if table_adam else None
if table_adam:      # the plain scatter + the separate table pass (the data-parallel route), for the scatter's own roofline record
    timer, plain_ms, pts_e = eager_pass(False)
    say(f"  with the tables' Adam step as a pass of its own (fused_table_adam=False): {plain_ms:.3f} ms/step; hg_levels_scatter {timer.us('hg_levels_scatter', n):.1f} us, "
        f"opt_adam_dense {timer.us('opt_adam_dense', n):.1f} us")

# ---- roofline records.
# Table scatter: the reference's 16 embedding_dense_backward calls (hash_encoding.py / network_hash.py:154-166) read-modify-write 8 rows x 8 B per point and
# level: 2 x 64 x 16 = 2048 B/point (SURVEY 8(d)'s backward formula with the 16 base tables in the place of the D codebook tables), + the feature gradients
# (128 B/point) and the position (12 B).  Priced against HBM; beside it the guide's float-atomic roof -- what a scatter that adds its 16 x 8 x 2 floats per point
# with global atomics cannot beat (1024 B of added bytes per point at 1.3 TB/s).
sc_s = timer.us("hg_levels_scatter", n) * 1e-6
sc_alg = 2048 + 128 + 12
# implemented: queue entries written and read once (16 B each: 4 per point on the 6 plain levels, 2 per run and (dy, dz) pair on the 10 merged ones -- counted as
# the upper bound 4 per point here), feature gradients + positions per level, every table row stored once by its owner
# Weight gradients.  One launch (default): field_bwd_wgrad = the MLP backward + the five products; what it must touch per point: the saved layer inputs (224 floats) +
# the encoder planes (32) read once, the backward's own inputs (upstream gradients, outputs, ReLU masks: 56 B), the feature gradients written (128 B).  Two launches
# (--two-launch): field_wgrad alone reads the layer inputs AND the pre-activation gradients field_bwd_trace wrote (480 floats).
wg_name = "field_bwd_wgrad" if fused else "field_wgrad"
wg_s = timer.us(wg_name, n) * 1e-6
wg_alg_flop = 2 * (64 * 32 + 16 * 64 + 64 * 32 + 64 * 64 + 16 * 64)          # per point: five products d x input^T
chain_mfma = 3 * (2 + 8 + 4 + 2 + 4)                                          # the backward chain per 32 points: 20 products of 32x32x16, 3 bf16 MFMAs each (split operands)
wg_issued_flop = (36 * 2 + (chain_mfma if fused else 0)) * 32768 / 32         # + 12 weight-gradient blocks x 2 K-steps x 3 MFMAs per 32 points
wg_bytes = (4 * (64 + 32 + 64 + 64) + 128 + 56 + 128) if fused else 4 * (64 + 64 + 32 + 32 + 16 + 16 + 64 + 64 + 64 + 64)      # 1208 | 1920 B/point
# HBM bytes from counters (a separate rocprofv3 --pmc run, tools/pmc_stage1.sh -> profiles/pmc_stage1.json), scaled from that run's point count to this one's
traffic_scatter = traffic_wgrad = traffic_source = None
try:
    pmc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_stage1.json")))
    k = pts_e / pmc["points_per_launch"]
    traffic_wgrad = pmc["k_field_bwd_wgrad" if fused else "k_field_wgrad"]["hbm_bytes_per_launch"] * k
    traffic_scatter = (pmc["k_level_entries"]["hbm_bytes_per_launch"] + pmc["k_scatter_binned"]["read_bytes"]) * k + pmc["k_scatter_binned"]["write_bytes"]      # (the owners store 16 whole tables whatever the point count)
    traffic_source = {"file": "profiles/pmc_stage1.json", "round": pmc.get("round"), "commit": pmc.get("commit"), "counters_taken_at_points": pmc["points_per_launch"],
                      "note": "NOT measured by this run: per-launch bytes of a separate rocprofv3 --pmc run, scaled linearly in the point count (the tables' 64 MiB store excepted)"}
except Exception:
    pass
out = {
    "scene": scene, "bound": BOUND, "cascades": int(m.cascade),
    "what": "stage-1 (clean model) training step, SURVEY 8(f) N3: 4096 rays of the scene (hotdog = S0, counter = S1), perturbed march, 16-level encoder, both MLPs with saved layer inputs, compositing, MSE, "
            "backward with the weight gradients inside (MFMA), 16-level owner-computes table scatter, Adam over 16 tables + both MLPs (torch.optim.Adam arithmetic), one hipGraph replay per step; "
            "update_extra_state every 16 steps between replays (inside the timed windows)",
    "ms_per_step": ms, "rays_per_s": n_rays / ms * 1e3, "rays": n_rays, "points_per_step": pts, "points_per_s": pts / ms * 1e3, "capacity_rows": capacity,
    "windows": win, "steps_per_window": steps, "grid_refresh_every": refresh, "recaptures": loop.recaptures, "capacity_overflow": bool(overflow), "loss_last": loss_last,
    "steps_trained": steps_done, "sparse_grid": sparse,
    "eager_one_stream_ms_per_step": eager_ms, "eager_points_per_step": pts_e,
    "kernels_us_per_step": {k: round(us, 1) for k, _, us in rows},
    "roofline_scatter": {
        "kernel": "hg_levels_scatter = k_level_entries (16 levels x chunks of 1024 points: queue entries sorted by slice in LDS) + k_scatter_binned (16 x 64 slice owners, LDS fixed-point sums, "
                  "every table row stored once: no zero fill, no global atomics, bit-reproducible)",
        "bound": "hbm", "avg_launch_s": sc_s, "points_per_launch": pts_e, "algorithmic_bytes_per_point": sc_alg,
        "achieved": pts_e * sc_alg / sc_s / 1e9 if sc_s else 0.0, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": pts_e * sc_alg / sc_s / HBM_PEAK if sc_s else 0.0,
        "traffic": traffic_scatter, "traffic_source": traffic_source, "frac_hbm_counters": (traffic_scatter / sc_s / HBM_PEAK) if (traffic_scatter and sc_s) else None,
        "basis": "the reference's algorithm: 16 x embedding_dense_backward = read-modify-write of 8 rows x 8 B per point and level (2048 B/point) + feature gradients (128) + position (12)",
        "float_atomic_roof": {"added_bytes_per_point": 1024, "peak_GBps": ATOMIC_PEAK / 1e9, "floor_s": pts_e * 1024 / ATOMIC_PEAK,
                              "this_launch_over_floor": (sc_s / (pts_e * 1024 / ATOMIC_PEAK)) if sc_s else None,
                              "note": "a global-float-atomic scatter cannot run faster than floor_s (MI355X_MICROARCH.md: ~1.3 TB/s of added bytes chip-wide); < 1 means the owner scheme beats that roof"},
        "plan_us_off_path": timer.us("hg_levels_plan", n)},
    "table_adam_in_owners": None if not table_adam else {
        "entry_point": "hg_levels_scatter_adam (the captured step's default in one process): k_level_entries + k_scatter_binned whose owners end with torch.optim.Adam's update of their "
                       "rows -- the 64 MiB of table gradients are neither written nor read back, the tables' pass leaves the step's tail; roofline_scatter above is the PLAIN scatter of a "
                       "second eager pass (the route a data-parallel step takes)",
        "avg_launch_s": fused_scatter_s, "bytes": "algorithmic 2188 B/point + parameters and both moments of 16 tables in and out (402.7 MB)",
        "achieved_GBps": (pts_e * sc_alg + 6 * 16 * 4194304) / fused_scatter_s / 1e9 if fused_scatter_s else None,
        "frac_hbm": (pts_e * sc_alg + 6 * 16 * 4194304) / fused_scatter_s / HBM_PEAK if fused_scatter_s else None,
        "replaces_us": {"hg_levels_scatter": round(timer.us("hg_levels_scatter", n), 1), "opt_adam_dense_over_18_tensors": round(timer.us("opt_adam_dense", n), 1)}},
    "roofline_wgrad": {
        "kernel": ("k_field_bwd_wgrad (the MLP backward with the weight gradients inside: each layer's pre-activation gradient transposed through wave-private LDS into MFMA operands over "
                   "the points, the layer inputs staged from memory two layers ahead, 12 accumulator blocks per wave in AGPRs over all its tiles, v_mfma_f32_32x32x16_bf16 on split hi + lo "
                   "operands, fp32 accumulate; one wave per SIMD) + k_wgrad_reduce (slabs in workgroup order)") if fused else
                  ("k_field_wgrad (split-K over the points: tiles of 32 points staged through LDS, one 32x32 product per wave, v_mfma_f32_32x32x16_bf16 on split hi + lo operands, "
                   "fp32 accumulate, slabs) + k_wgrad_reduce (fixed order)"),
        "entry_point": wg_name,
        "bound": "hbm", "avg_launch_s": wg_s, "points_per_launch": pts_e, "algorithmic_bytes_per_point": wg_bytes,
        "achieved": pts_e * wg_bytes / wg_s / 1e9 if wg_s else 0.0, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": pts_e * wg_bytes / wg_s / HBM_PEAK if wg_s else 0.0,
        "traffic": traffic_wgrad, "traffic_source": traffic_source, "frac_hbm_counters": (traffic_wgrad / wg_s / HBM_PEAK) if (traffic_wgrad and wg_s) else None,
        "basis": ("every saved layer input row and encoder plane read exactly once (256 floats per point), the backward's inputs (56 B) read, the feature gradients (128 B) written; the "
                  "pre-activation gradients never leave the chip.  What binds the launch is neither roof: one wave per SIMD (192 accumulator registers, 120 KiB of LDS per workgroup) "
                  "runs the backward chain's dependent MFMA -> VALU -> LDS sequences unhidden (tools/_ab_fused.py: 163 us without the loads, 102 us without loads and products, at 673 k points)")
                 if fused else "every saved layer input and pre-activation gradient row read exactly once (480 floats per point); the products are K = points reductions, 10 FLOP per byte: HBM-bound",
        "replaces_us_per_step": None if not fused else {"note": "field_bwd_trace_rows + field_wgrad of the two-launch route (--two-launch), round-5 record at 620 k points", "us": 157.9 + 298.4},
        "mfma": {"algorithmic_flop_per_point": wg_alg_flop, "issued_flop_per_point": wg_issued_flop,
                 "achieved_TFLOPs_issued": pts_e * wg_issued_flop / wg_s / 1e12 if wg_s else 0.0, "frac_of_dense_bf16_peak": pts_e * wg_issued_flop / wg_s / MFMA_PEAK_BF16 if wg_s else 0.0}},
    # (--rccl1) what a data-parallel rank exchanges per step: levels 5..15 dense + the static live rows of levels 0..4 + both MLPs (stage1.live_rows); dense would be 64 MiB + 40 KB
    "exchange": {"bytes_per_step": loop.bytes_exchanged_per_step, "dense_bytes_per_step": (16 * (1 << 19) * 2 + 3072 + 7168) * 4, "segments": n_segments,
                 "ring_bytes_per_gpu_at_8_ranks": loop.bytes_exchanged_per_step * 2 * 7 / 8} if "--rccl1" in flags else None,
    "parity": "tests/test_gpu_stage1.py: all-parameter gradients (16 levels) vs the oracle's autograd, weight gradients vs fp64 products, captured == eager, 200 steps tracked by the CPU oracle, "
              "grid-refresh cadence, two-rank exchange == single-process gradient",
}
if as_json:
    print(json.dumps(out), flush=True)
