#!/usr/bin/env python
"""Phase timing inside the decoder's conv kernels: builds a private copy of libnerfsig with -DNSIG_DEC_TIMING (here, in the
build container:  python tools/dec_timing.py --build), then on the GPU box runs one forward+backward and prints the
100 MHz wall-clock stamps of workgroup (0,0):  python tools/dec_timing.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "_build", "libnerfsig_timing.so")

if "--build" in sys.argv:
    srcs = sorted(os.path.join(ROOT, "nerf_signature_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "nerf_signature_amd", "csrc")) if f.endswith(".hip"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
                           "-DNSIG_DEC_TIMING", "-shared", "-o", LIB, *srcs])
    print(LIB)
    sys.exit(0)

sys.path.insert(0, ROOT)
os.environ["NERFSIG_LIB"] = LIB
import torch
from nerf_signature_amd import _native as nv
from nerf_signature_amd.hidden_models import get_hidden_decoder_multi_views

assert nv.load()._name == LIB, nv.load()._name
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 12, 12)))
dec = get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=3, channels=64).cuda()
img = torch.randn(B, 3, H, W, device="cuda", requires_grad=True)
for _ in range(3):
    dec(img).sum().backward()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 48)()
fn = nv.load().dec_timing_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(out) == 0
names = ["start", "A loads issued", "stats combined", "staged", "barrier", "mfma done", "epilogue done", "partials issued", "inputs issued"]
for m, mode in enumerate(("fwd", "dgrad", "dgrad_img")):
    st = list(out[m * 16:(m + 1) * 16])
    print(mode, " ".join(f"{names[k]}={(st[k] - st[0]) * 10:d}ns" for k in (7, 8, 1, 2, 3, 4, 5, 6) if st[k]))

