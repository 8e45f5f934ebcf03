#!/usr/bin/env python
"""Phase times inside k_level_entries (instrumented build: tools/build_variant.sh enttiming -DNSIG_ENT_TIMING)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NERFSIG_LIB"] = os.path.join(ROOT, "tools", "_build", "libnerfsig_enttiming.so")
sys.argv = [sys.argv[0], "content", "--no-overlap", "--steps", "32", "--windows", "2"]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "tools", "stage1_bench.py"), run_name="__main__")
except SystemExit:
    pass
from nerf_signature_amd import _native as nv
out = (ctypes.c_ulonglong * 144)()
fn = nv.load().level_entries_phase_ticks
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(out, 0) == 0
names = ["inputs", "cells+ranks", "run sums", "barrier", "prefix+barrier", "staging", "barrier", "copy-out+max"]
print("k_level_entries, mean us per workgroup and phase (thread 0's clock), by level:")
print("level  " + "  ".join(f"{n:>14s}" for n in names) + "           total")
for l in range(16):
    n = max(out[9 * l + 8], 1)
    v = [out[9 * l + i] * 0.01 / n for i in range(8)]
    print(f"{l:5d}  " + "  ".join(f"{x:14.2f}" for x in v) + f"  {sum(v):14.2f}")
