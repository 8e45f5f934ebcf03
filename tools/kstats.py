#!/usr/bin/env python
"""Print the top kernels of a rocprofv3 --kernel-trace --stats run (the *_kernel_stats.csv under a directory)."""
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{f}: {len(rows)} kernels, total {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.3f} ms/step over {steps:g} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{r['Name'][:70]:70s} n={int(r['Calls']):5d} avg={float(r['AverageNs']) / 1e3:8.1f}us per-step={float(r['TotalDurationNs']) / 1e6 / steps:7.3f}ms {float(r['Percentage']):5.1f}%")
