#!/bin/bash
# usage (GPU box): tools/prof_round.sh <tag>  -> gpurun_out/<tag>_*: the bench line, rocprofv3 kernel stats of the same command, the
# encoder / MLP launches split by size, and the timeline of one graph replay
tag=$1
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out      # (an earlier run of the same tag would otherwise leave its traces beside the new ones)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --windows 1 > $out.log 2>&1) || exit 1
python tools/kstats.py $out 30 40 > gpurun_out/${tag}_summary.txt
python - $out >> gpurun_out/${tag}_summary.txt <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print("\n# dominant kernels split by launch size (grid x workgroup), durations in us")
for key in ("k_encode_planes", "k_field_fwd", "k_field_bwd", "k_scatter_binned", "k_codebook_adam_sel", "k_march_index"):
    by = collections.defaultdict(list)
    for r in rows:
        if key in r["Kernel_Name"]:
            by[(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{key:22s} grid={k[0]:>9s} wg={k[1]:>5s} n={len(v):4d} avg={sum(v)/len(v):8.1f} min={min(v):8.1f} max={max(v):8.1f}")
PY
python - $out >> gpurun_out/${tag}_summary.txt <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
enc = [r for r in rows if "k_encode_planes" in r["Kernel_Name"]]
big = max(int(r.get("Grid_Size_X") or r.get("Grid_Size")) for r in enc if int(r.get("Grid_Size_X") or r.get("Grid_Size")) < 12000000)
seq = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in enc if int(r.get("Grid_Size_X") or r.get("Grid_Size")) == big]
print(f"\n# k_encode_planes, the block render's launch (grid {big}), every launch of the run in order, us:")
print("#   prepare (sizing + 3 warm-up) | 25 graph replays (beside the content render's kernels: inflated) | the 5 EAGER launches bench.py's HIP events bracket | the variant's set-up")
print("  " + " ".join(f"{v:.0f}" for v in seq))
if len(seq) >= 34:
    eager = seq[29:34]
    print(f"#   eager launches {eager}: avg {sum(eager) / 5:.1f} us  (compare roofline.avg_launch_s of the bench line of the same box)")
PY
python tools/timeline.py $out k_adam_prepare ${TL_BACK:-8} > gpurun_out/${tag}_timeline_fixed_blocks_variant.txt   # (the run ends with the variant's replays)
python tools/timeline.py $out k_adam_prepare ${TL_HEAD:-45} > gpurun_out/${tag}_timeline_headline.txt
cp $(ls $out/*/*_kernel_stats.csv | tail -1) gpurun_out/${tag}_kernel_stats.csv
