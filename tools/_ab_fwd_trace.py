#!/usr/bin/env python
"""Same-box A/B of field_fwd_trace's two launches (mlp_set_pipelined bit 0: k_field_fwd_trace | the generic kernel's trace variant) at the bench step's point count.

    python tools/_ab_fwd_trace.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import _native as nv, fieldops as fo, stage1
from nerf_signature_amd.stage1 import CleanNeRFNetwork

M = int(sys.argv[1]) if len(sys.argv) > 1 else 673_478
torch.manual_seed(0)
m = CleanNeRFNetwork(bound=1.0, cuda_ray=False).cuda()
pts = torch.rand(M, 3, device="cuda") * 2 - 1
dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda"), dim=-1)
tr = stage1._Traces(M, pts.device, with_grads=False)
packed = fo.pack_weights(m.sigma_net.params, m.color_net.params)
base = nv.ptr_array([t.detach() for t in m.encoder.tables()])
nv.call("hg_encode_planes", nv.ptr(pts), M, 1.0, base, None, nv.ptr(tr.planes), nv.stream())
before = nv.fn("mlp_get_pipelined")()
for rnd in range(3):
    for mask, name in ((3, "k_field_fwd_trace (inputs one tile ahead)"), (2, "k_field_fwd<Bf16x3, 1, true> (plain loop)")):
        nv.call("mlp_set_pipelined", mask)
        args = (nv.ptr(pts), nv.ptr(dirs), M, 1.0, base, nv.ptr(packed), nv.ptr(tr.planes), nv.ptr(tr.sig), nv.ptr(tr.rgb), nv.ptr(tr.masks), *[nv.ptr(a) for a in tr.act], nv.stream())
        for _ in range(3):
            nv.call("field_fwd_trace", *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            nv.call("field_fwd_trace", *args)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {rnd}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us  {name}", flush=True)
nv.call("mlp_set_pipelined", before)
