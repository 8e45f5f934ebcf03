#!/bin/bash
# per-level owner times of the stage-1 table scatter (variant built with -DNSIG_LEVELS_SPLIT) + phase shares inside k_level_entries (-DNSIG_ENT_TIMING)
cd $GRAFT_REPO_ROOT
python tools/entries_timing.py 2>&1 | grep -E "captured step|mean us per workgroup|^level|^ +[0-9]+ "
out=$GRAFT_REPO_ROOT/gpurun_out/prof_levels
rm -rf $out
(cd /tmp && export TMPDIR=/tmp && NERFSIG_LIB=$GRAFT_REPO_ROOT/tools/_build/libnerfsig_split.so rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/stage1_bench.py content --windows 2 --steps 32 > $out.log 2>&1)
python tools/levels_trace.py $out
python - $out <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    tail = v[-40:]
    print(f"{k:62s} n={len(v):5d} avg of last 40: {sum(tail)/len(tail):8.1f} us")
PY
