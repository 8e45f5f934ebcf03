#!/usr/bin/env python
"""A/B of build variants of field_bwd_wgrad (csrc/stage1_fused.hip) alone, at the bench step's point count.

    python tools/_ab_fused.py [M] -- VARIANT_FLAGS ...     e.g.  -- "" "-DNSIG_FUSED_NOLOAD" "-DNSIG_FUSED_NOPROD"

Each variant: the library rebuilt into a scratch directory with the extra flags on stage1_fused.hip only, loaded by a child process (NERFSIG_LIB), which runs the
forward trace on M random points once and times 20 launches of the fused backward with HIP events on the launch stream."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
from nerf_signature_amd import _native as nv, fieldops as fo, stage1
from nerf_signature_amd.stage1 import CleanNeRFNetwork
M = %d
torch.manual_seed(0)
m = CleanNeRFNetwork(bound=1.0, cuda_ray=False).cuda()
pts = (torch.rand(M, 3, device="cuda") * 2 - 1)
dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda"), dim=-1)
gs, gc = torch.randn(M, device="cuda") * 1e-4, torch.randn(M, 3, device="cuda") * 1e-4
tr = stage1._Traces(M, pts.device)
packed = fo.pack_weights(m.sigma_net.params, m.color_net.params)
base = nv.ptr_array([t.detach() for t in m.encoder.tables()])
stage1._forward_trace(tr, pts, dirs, 1.0, base, packed)
g_sp, g_cp = torch.empty(3072, device="cuda"), torch.empty(7168, device="cuda")
for _ in range(3):
    stage1._backward_trace(tr, gs, gc, packed, g_sp, g_cp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    stage1._backward_trace(tr, gs, gc, packed, g_sp, g_cp)
e1.record(); torch.cuda.synchronize()
print("%%.1f us per launch (incl. the slab reduction), |g_sp| %%.4e" %% (e0.elapsed_time(e1) / 20 * 1e3, float(g_sp.abs().sum())))
'''


def main():
    argv = sys.argv[1:]
    split = argv.index("--") if "--" in argv else len(argv)
    M = int(argv[0]) if split > 0 else 673_478
    variants = argv[split + 1:] or [""]
    from nerf_signature_amd import build as b
    b.build()
    for i, v in enumerate(variants):
        d = f"/tmp/ab_fused_{i}"
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, "stage1_fused.o")
        src = os.path.join(b.CSRC, "stage1_fused.hip")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *b.flags_for(src), *v.split(), "-c", src, "-o", obj])
        objs = [os.path.join(b.LIB_DIR, os.path.basename(s)[:-4] + ".o") for s in b.sources() if not s.endswith("stage1_fused.hip")] + [obj]
        lib = os.path.join(d, "libnerfsig.so")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", lib, *objs])
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, M)], env={**os.environ, "NERFSIG_LIB": lib}, capture_output=True, text=True, timeout=300)
        print(f"[{v or 'as built'}] {out.stdout.strip() or out.stderr.strip()[-400:]}", flush=True)


if __name__ == "__main__":
    main()
