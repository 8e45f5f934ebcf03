#!/usr/bin/env python
"""gpurun_out/<tag>_stage1_pmc.txt (tools/pmc_stage1.sh) -> profiles/pmc_stage1.json: per-kernel HBM bytes per launch of the stage-1 step's kernels.

    python tools/pmc_stage1_json.py gpurun_out/<tag>_stage1_pmc.txt <commit> > profiles/pmc_stage1.json

Units and corrections as MI355X_MICROARCH.md's HBM section prescribes (and profiles/pmc_traffic.json calibrated: 1.003 on the 836 MiB Adam pass): FETCH_SIZE counts KiB at 64 B
per request on gfx950 where the request is 128 B -> x 1024 x 2; WRITE_SIZE KiB x 1024.  Each kernel's row is the mean over the last 10 launches of the run's eager pass (one
stream, nothing beside the kernel)."""
import json
import re
import sys

KERNELS = {"k_field_bwd_wgrad": "k_field_bwd_wgrad", "k_field_wgrad": "k_field_wgrad", "k_level_entries": "k_level_entries", "k_scatter_binned": "k_scatter_binned",
           "k_levels_count": "k_levels_count", "k_field_fwd<Bf16x3, 1, true>": "k_field_fwd_trace", "k_field_fwd_trace": "k_field_fwd_trace", "k_field_fwd_trace<float>": "k_field_fwd_trace", "k_field_bwd_wgrad<float>": "k_field_bwd_wgrad", "k_field_bwd<Bf16x3, true>": "k_field_bwd_trace",
           "k_encode_planes": "k_encode_planes", "k_encode_planes<false>": "k_encode_planes", "k_wgrad_reduce": "k_wgrad_reduce"}


def main():
    path, commit = sys.argv[1], sys.argv[2]
    rows, points = {}, None
    for line in open(path):
        m = re.search(r"eagerly on one stream: [\d.]+ ms/step, (\d+) points/step", line)
        if m:
            points = int(m.group(1))
            continue
        m = re.match(r"(\S.*?)\s+grid=(\d+)\s+n=(\d+)\s+(.*)$", line)
        if not m or m.group(1) not in KERNELS or int(m.group(3)) != 10:      # (n = 10: the eager pass' launches; the set-up's launches of the same kernel have other counts)
            continue
        vals = {k: float(v) for k, v in (kv.split("=") for kv in m.group(4).split())}
        vals["_grid"] = int(m.group(2))
        if KERNELS[m.group(1)] not in rows or rows[KERNELS[m.group(1)]]["_grid"] < vals["_grid"]:      # (a kernel launched at two sizes: the bench step's -- the larger -- launch)
            rows[KERNELS[m.group(1)]] = vals
    out = {"round": int(sys.argv[3]) if len(sys.argv) > 3 else 6, "commit": commit, "points_per_launch": points,
           "source": "rocprofv3 --pmc, one counter group per run, kernel trace only (tools/pmc_stage1.sh) over tools/stage1_bench.py; means over the last 10 launches (the eager pass on one stream)",
           "method": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 tallies a 128-byte request at 64 B; TCC_BUBBLE does not count on this part: profiles/pmc_traffic.json calibration 1.003), WRITE_SIZE [KiB] x 1024"}
    per_point = {}
    for name, v in rows.items():
        rd, wr = v.get("FETCH_SIZE", 0.0) * 1024 * 2, v.get("WRITE_SIZE", 0.0) * 1024
        out[name] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT"), "lds_instructions": v.get("SQ_INSTS_LDS"),
                     "l1_to_l2_requests": v.get("TCP_TCC_READ_REQ_sum"), "valu_instructions": v.get("SQ_INSTS_VALU"), "mfma_instructions": v.get("SQ_INSTS_MFMA"),
                     "wave_cycles": v.get("SQ_WAVE_CYCLES"), "wait_any_cycles": v.get("SQ_WAIT_ANY"), "active_inst_any_cycles": v.get("SQ_ACTIVE_INST_ANY")}
        if points:
            per_point[name] = (rd + wr) / points
    out["per_point"] = per_point
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
