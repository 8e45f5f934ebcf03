#!/usr/bin/env python
"""Print the kernel timeline of the last full step of a rocprofv3 --kernel-trace run: start (us, relative), duration, queue, kernel."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True))[-1]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_march_index'
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
