#!/usr/bin/env python
"""profiles/pmc_traffic.json from a tools/pmc_step.sh run: per-launch HBM-side bytes and the L1/L2 counters of the step's dominant kernels, with the
round and the commit the counters were taken at (bench.py prints them as roofline.traffic_source).
    python tools/pmc_traffic.py gpurun_out/r04_pmc.txt 4 > profiles/pmc_traffic.json
Byte conversion (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB and equals TCC_EA0_RDREQ x 64 B; on gfx950 a 128-byte request is tallied at 64 B,
so read bytes = FETCH_SIZE x 1024 x (1 + share of 128-byte requests) -- the share is measured (TCC_EA0_RDREQ_128B / TCC_EA0_RDREQ) when the run has it,
else taken as 1 (round 1 calibrated these kernels at 0.999 on k_codebook_presum's exactly known 128 MiB).  WRITE_SIZE x 1024 is exact."""
import json
import re
import subprocess
import sys

path, rnd = sys.argv[1], int(sys.argv[2])
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
rows = {}
tail = ""
for line in open(path):
    if line.startswith("PMC_STEP_END"):
        tail = line.strip()
        continue
    name = line[:58].strip()
    kv = dict(p.split("=", 1) for p in line[58:].split() if "=" in p)
    if "grid" not in kv:
        m = re.search(r"grid=\s*(\d+)", line)
        kv["grid"] = m.group(1) if m else "?"
    rows[(name, kv.pop("grid"))] = {k: float(v) for k, v in kv.items() if k != "n"}


def pick(name, grid=None):
    c = [(k, v) for k, v in rows.items() if k[0] == name and (grid is None or k[1] == str(grid))]
    return c[0][1] if c else None


def traffic(c):
    if c is None:
        return None
    # the 128-byte read requests: TCC_BUBBLE on gfx950 (counter_defs.yaml: "Number of 128-byte read requests sent to EA").  Measured where the run has it and it
    # counts (> 0); otherwise ASSUMED 1.0 (round 1 calibrated these kernels at 0.999 on k_codebook_presum's exactly known 128 MiB) -- and the record says which.
    big = c.get("TCC_BUBBLE_sum") or c.get("TCC_EA0_RDREQ_128B_sum")
    # (on this gfx950 TCC_BUBBLE reads zero -- or a few counts of noise -- for every kernel, the 836 MiB streaming pass included: it does not count there.  "Measured" only
    #  when it accounts for at least 1 % of the requests; the calibration entry below says how good the assumption is.)
    measured = bool(big) and bool(c.get("TCC_EA0_RDREQ_sum")) and big >= 0.01 * c["TCC_EA0_RDREQ_sum"]
    share = min(1.0, big / c["TCC_EA0_RDREQ_sum"]) if measured else 1.0
    # measured: bytes straight from the request counters (128-byte, 32-byte, the rest 64-byte); assumed: the guide's correction of FETCH_SIZE (= RDREQ x 64 B)
    rd = (big * 128 + (c["TCC_EA0_RDREQ_sum"] - big - c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) * 64 + c.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32) if measured else c["FETCH_SIZE"] * 1024 * (1 + share)
    wr = c["WRITE_SIZE"] * 1024
    out = {"FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"], "share_of_128B_read_requests": share,
           "share_source": "measured (TCC_BUBBLE_sum / TCC_EA0_RDREQ_sum; read bytes = 128 x BUBBLE + 32 x RDREQ_32B + 64 x the rest)" if measured else "ASSUMED 1.0 (TCC_BUBBLE_sum absent or zero in this run)",
           "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    for k in ("TCC_EA0_RDREQ_sum", "TCC_BUBBLE_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_ATOMIC_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TA_TA_BUSY_sum",
              "TA_FLAT_READ_WAVEFRONTS_sum", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
              "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
        if k in c:
            out[k] = c[k]
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    return out


enc_rows = sorted(((k, v) for k, v in rows.items() if k[0].startswith("k_encode_planes") and "codebook" not in k[0]), key=lambda kv: -int(kv[0][1]))
# the block render's launch: 8 workgroup slots x 1,290,240 rows (the bench workload); larger grids belong to the set-up's clean renders
enc_block = next((v for k, v in enc_rows if int(k[1]) == 8 * 1290240), None)
enc_content = next((v for k, v in enc_rows if int(k[1]) < 8 * 1290240), None)
doc = {
    "round": rnd, "commit": commit,
    "source": f"rocprofv3 --pmc (one counter group per run, --kernel-trace only: tools/pmc_step.sh) over tools/pmc_step.py: the bench workload's training step issued eagerly on ONE "
              f"stream in the captured step's order, so every kernel sees the cache state it has in the replayed step -- the block render's hash gather at the head of the step, "
              f"right behind the optimiser's 836 MiB stream and the table warm-up pass; per-launch means over the last 6 steps; MI355X, round {rnd}, commit {commit}. {tail}",
    "method": "MI355X_MICROARCH.md 'HBM': FETCH_SIZE [KiB] = TCC_EA0_RDREQ x 64 B; a 128-byte request is tallied at 64 B on gfx950, so read bytes = FETCH_SIZE x 1024 x (1 + share of 128-B "
              "requests); WRITE_SIZE x 1024 exact.  Infinity-Cache hits are counted in both (fabric-side counters).",
    "k_encode_planes": dict(traffic(enc_block) or {}, TCP_TCC_READ_REQ=(enc_block or {}).get("TCP_TCC_READ_REQ_sum"), TCP_TOTAL_CACHE_ACCESSES=(enc_block or {}).get("TCP_TOTAL_CACHE_ACCESSES_sum"),
                            launch="block render, 1,290,240 rows, head of the step"),
    "k_encode_planes_content_launch": traffic(enc_content),
    "k_field_fwd_train": traffic(pick("k_field_fwd_train<F16, true> [block render]") or pick("k_field_fwd_train<F16, false> [block render]") or pick("k_field_fwd_train<F16> [block render]")),
    "k_field_bwd_train": traffic(pick("k_field_bwd_train<F16> [block render]")),
    "k_scatter_binned": traffic(pick("k_scatter_binned")),
    "k_scatter_merge": traffic(pick("k_scatter_merge")),      # (round 6: the exact merge of the owners' replicas)
    "k_codebook_adam_sel_next": traffic(pick("k_codebook_adam_sel<true, true>")),
    "k_warm_tables": traffic(pick("k_warm_tables")),
}
doc["k_encode_planes_hbm_bytes_per_launch"] = doc["k_encode_planes"].get("hbm_bytes_per_launch")
m = re.search(r"block render (\d+)", tail)      # the point count the block render's launch had when the counters were taken: bench.py scales the per-launch counts to its own launch
doc["k_encode_planes_points_per_launch"] = int(m.group(1)) if m else None
adam = doc.get("k_codebook_adam_sel_next") or {}
if adam.get("hbm_bytes_per_launch"):
    exact = 836 * 2 ** 20      # G once + (param, exp_avg, exp_avg_sq) of 32 tables read and written + partner tables + S (bench.py step_bytes): 836 MiB
    doc["calibration"] = {"kernel": "k_codebook_adam_sel<NEXT>: moves exactly 836 MiB per launch", "bytes_by_this_method": adam["hbm_bytes_per_launch"], "exact_bytes": exact,
                          "ratio": adam["hbm_bytes_per_launch"] / exact}
print(json.dumps(doc, indent=1))
