#!/usr/bin/env python
"""Times the HiDDeN decoder alone (forward, forward + backward) on the bench shape: eagerly and replayed from a hipGraph."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd.hidden_models import get_hidden_decoder_multi_views

B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32, 12, 12)))
torch.manual_seed(0)
dec = get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=3, channels=64).cuda()
img = torch.rand(B, H, W, 3, device="cuda", requires_grad=True)
gout = torch.randn(B, 1, device="cuda")
params = [img] + list(dec.parameters())


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def fwd():
    with torch.no_grad():
        return dec.decode_rendered(img)[0]


def fwd_bwd():
    out, _ = dec.decode_rendered(img)
    torch.autograd.grad(out, params, gout, allow_unused=True)


def graphed(fn):
    """The same call replayed from a hipGraph: launch gaps as in the captured training step, no host time."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


for _ in range(2):
    print(f"{B}x{H}x{W}: forward {timeit(fwd):7.1f} us, forward+backward {timeit(fwd_bwd):7.1f} us (eager) | "
          f"forward {timeit(graphed(fwd)):7.1f} us, forward+backward {timeit(graphed(fwd_bwd)):7.1f} us (hipGraph replay)")
