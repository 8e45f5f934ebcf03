#!/usr/bin/env python
"""Per-kernel mean of the counters of one rocprofv3 --pmc run (counter_collection.csv under the directory)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else "nsig::"
files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
if not files:
    sys.exit(f"no counter_collection.csv under {d}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name") or r.get("Kernel Name") or ""
        if only not in name:
            continue
        short = name.split("(")[0][-40:]
        acc[short][r["Counter_Name"]].append((r.get("Dispatch_Id"), float(r["Counter_Value"])))
for k, cs in acc.items():
    parts = []
    for c, vals in sorted(cs.items()):
        per = collections.defaultdict(float)
        for disp, v in vals:
            per[disp] += v
        xs = list(per.values())
        parts.append(f"{c}={sum(xs) / len(xs):.4g}")
    print(f"{k:40s} n={len(next(iter(cs.values())))} " + " ".join(parts))
