#!/usr/bin/env python
"""Phase stamps inside k_scatter_sliced (instrumented build: python tools/dec_timing.py --build first)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NERFSIG_LIB"] = os.path.join(ROOT, "tools", "_build", "libnerfsig_timing.so")
import torch
from nerf_signature_amd import _native as nv
from nerf_signature_amd import fieldops as fo

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1290240
torch.manual_seed(0)
cell = torch.randint(0, 2048, (M, 3), device="cuda", dtype=torch.int32)
rec = torch.zeros(M, 8, device="cuda")
rec.view(torch.int32)[:, 0] = cell[:, 0] | (cell[:, 1] << 16)
rec.view(torch.int32)[:, 1] = cell[:, 2]
rec[:, 2:5] = torch.rand(M, 3, device="cuda")
rec[:, 5:7] = torch.randn(M, 2, device="cuda")
G = torch.zeros(1 << 19, 2, device="cuda")
for _ in range(3):
    fo.codebook_scatter_sliced(rec, G)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
fn = nv.load().scatter_timing_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out) == 0
names = ["start", "LDS zeroed", "scan done (wave 0)", "scan done (all)", "flushed"]
print(" ".join(f"{names[k]}={(out[k] - out[0]) * 10}ns" for k in range(1, 5)))
