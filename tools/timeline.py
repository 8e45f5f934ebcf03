#!/usr/bin/env python
"""Print the kernel timeline of the last complete step of a rocprofv3 --kernel-trace run: start offset, duration, gap to the
previous kernel's end, name.  A step is delimited by consecutive launches of the marker kernel (default k_adam_prepare)."""
import csv
import glob
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_adam_prepare"
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # 1 = last step, 2 = the one before, ...
a, b = idx[-1 - back] + 1, idx[-back] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f}us  dur={(e - s) / 1e3:7.1f}  gap={(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, {b - a} kernels")
