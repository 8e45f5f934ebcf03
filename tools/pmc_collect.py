#!/usr/bin/env python
"""Per-(kernel, grid size) means of the counters of rocprofv3 --pmc runs (one directory per counter group), restricted to the LAST `keep`
launches of each (kernel, grid) -- the eager steps of tools/pmc_step.py, not its set-up.   usage: pmc_collect.py <keep> <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

keep = int(sys.argv[1])
out = collections.defaultdict(dict)
for d in sys.argv[2:]:
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # key -> counter -> dispatch -> value
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or ""
            if "nsig::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("nsig::", "")
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
            per[(short, grid)][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for key, cs in per.items():
        # persistent-grid kernels are launched twice per step with the SAME grid (block render first, then content render): split by position
        twice = key[0].startswith(("k_field_fwd_train", "k_field_bwd_train"))
        for c, disp in cs.items():
            if twice:
                ids = sorted(disp)[-2 * keep:]
                halves = sorted((ids[0::2], ids[1::2]), key=lambda sub: -sum(disp[i] for i in sub))     # the block render's launch is the bigger one in every counter
                for tag, sub in zip(("block render", "content render"), halves):
                    k2 = (key[0] + " [" + tag + "]", key[1])
                    out[k2][c] = sum(disp[i] for i in sub) / len(sub)
                    out[k2]["_launches_averaged"] = len(sub)
                continue
            ids = sorted(disp)[-keep:]
            out[key][c] = sum(disp[i] for i in ids) / len(ids)
            out[key]["_launches_averaged"] = len(ids)
for (k, g), cs in sorted(out.items(), key=lambda kv: (kv[0][0], -int(kv[0][1]) if kv[0][1].isdigit() else 0)):
    print(f"{k[:58]:58s} grid={g:<9s} n={cs.pop('_launches_averaged')} " + " ".join(f"{c}={v:.5g}" for c, v in sorted(cs.items())))
