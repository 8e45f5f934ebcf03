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
out = (ctypes.c_ulonglong * 16)()
fn = nv.load().level_entries_phase_ticks
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(out, 0) == 0
names = ["inputs", "cells+ranks", "run sums", "barrier", "prefix+barrier", "staging", "barrier", "copy-out+max"]
for k, kind in enumerate(("merged levels", "plain levels")):
    tot = sum(out[8 * k + i] for i in range(8))
    print(kind, "share of workgroup time:", "  ".join(f"{names[i]} {100.0 * out[8 * k + i] / max(tot, 1):.0f}%" for i in range(8)), f"  (total {tot * 10 / 1e3:.0f} us over all workgroups and steps)")
