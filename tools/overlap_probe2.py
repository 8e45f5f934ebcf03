#!/usr/bin/env python
"""Where could the block render's base-level hash gather (k_encode_planes without the codebook level: nothing it reads is changed by a
training step) hide?  Times it alone and side by side -- two streams -- with the step's other phases: the codebook Adam (HBM stream), the
decoder's forward + backward chain (dependent small launches), the MLP backward + scatter.  NERFSIG_ENC_PER_SLOT throttles the encoder's grid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_signature_amd import _native as nv
from nerf_signature_amd import fieldops as fo
from nerf_signature_amd import synthetic
from nerf_signature_amd.network import NeRFNetwork

dev = torch.device("cuda")
torch.manual_seed(0)
D, T = 32, 1 << 19
model = NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=D, n_views=1)
synthetic.init_model(model, "hotdog")
model.to(dev).train()
bo, bd = synthetic.block_rays("hotdog", dev)
rec = model.fix_rays(bo, bd, 0, 1024)
xyzs, M = rec["xyzs"], rec["xyzs"].shape[0]
base = model.encoder.tables()
base_ptrs = nv.ptr_array([t.detach() for t in base])
planes = torch.empty(int(nv.fn("hg_planes_bytes")(M)), dtype=torch.uint8, device=dev)

tabs = [torch.randn(T, 2, device=dev) * 1e-4 for _ in range(2 * D)]
m1 = [torch.zeros(T, 2, device=dev) for _ in range(2 * D)]
m2 = [torch.zeros(T, 2, device=dev) for _ in range(2 * D)]
steps = [torch.zeros((), device=dev) for _ in range(2 * D)]
G = torch.randn(T, 2, device=dev) * 1e-3
msg = torch.randint(0, 2, (D,), device=dev).float()
lr = torch.tensor(1e-2, device=dev)
scratch = torch.empty(2 * D, device=dev)
arrs = [nv.ptr_array(x) for x in (tabs, m1, m2, steps)]

dec = model.msg_decoder
img = torch.rand(D, 12, 12, 3, device=dev, requires_grad=True)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def encode(st):
    nv.call("hg_encode_planes", nv.ptr(xyzs), M, 1.0, base_ptrs, None, nv.ptr(planes), st.cuda_stream)


def adam(st):
    nv.call("opt_codebook_adam_sel", nv.ptr(G), *arrs, nv.ptr(msg), D, nv.ptr(lr), 0.9, 0.99, 1e-15, 1.0, nv.ptr(scratch), st.cuda_stream)


def decoder(st):
    with torch.cuda.stream(st):
        out, _ = dec.decode_rendered(img)
        out.sum().backward()


# MLP backward + scatter of the block render (through the kept plan)
S = fo.codebook_presum(fo.select_tables(model.msg_encoder.tables(), tuple(int(v) for v in msg.tolist())))
packed = model._packed()
sig, rgb, _, masks = fo.field_forward(xyzs, rec["dirs"], 1.0, base, S, packed, want_masks=True, fixed=rec["fixed"])
gs, gr = torch.randn_like(sig) * 1e-3, torch.randn_like(rgb) * 1e-3
Gs = torch.zeros(T, 2, device=dev)


def backward(st):
    with torch.cuda.stream(st):
        fo.field_backward_planned(xyzs, 1.0, gs, gr, sig, rgb, masks, packed, rec["fixed"].plan, Gs)


def forward(st):
    with torch.cuda.stream(st):
        fo.field_forward(xyzs, rec["dirs"], 1.0, base, S, packed, want_masks=True, fixed=rec["fixed"])


def timed(fn, reps=30):
    """Mean time of one fork/join of the two streams, with no host synchronisation between the repetitions (the GPU stays at its clocks)."""
    s0 = torch.cuda.current_stream()

    def rep():
        s1.wait_stream(s0); s2.wait_stream(s0)
        fn()
        s0.wait_stream(s1); s0.wait_stream(s2)

    for _ in range(5):
        rep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rep()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"ENC_PER_SLOT={os.environ.get('NERFSIG_ENC_PER_SLOT', 'default')}  points {M}")
t_enc = timed(lambda: encode(s1))
print(f"encoder (16 base levels) alone {t_enc:.1f} us")
for name, fn in (("adam", adam), ("decoder fwd+bwd", decoder), ("mlp forward (kept planes)", forward), ("mlp backward + scatter", backward)):
    alone = timed(lambda: fn(s2))
    both = timed(lambda: (encode(s1), fn(s2)))
    rev = timed(lambda: (fn(s2), encode(s1)))
    print(f"{name:28s} alone {alone:7.1f} us | beside the encoder {both:7.1f} us (issued second) {rev:7.1f} us (issued first) | one after the other {alone + t_enc:7.1f}")

# ---- the same question for the decoder chain inside a captured graph (no host in the loop): decoder alone, encoder alone, both as two
# branches of one graph
def capture(fn):
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        g.capture_begin()
        s1.wait_stream(cs); s2.wait_stream(cs)
        fn()
        cs.wait_stream(s1); cs.wait_stream(s2)
        g.capture_end()
    torch.cuda.current_stream().wait_stream(cs)
    return g


def replay_time(g, reps=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


img.grad = None
for _ in range(3):
    decoder(s2)
torch.cuda.synchronize()
g_dec = capture(lambda: decoder(s2))
g_enc = capture(lambda: encode(s1))
g_both = capture(lambda: (encode(s1), decoder(s2)))
g_both_r = capture(lambda: (decoder(s2), encode(s1)))
print(f"captured: decoder fwd+bwd alone {replay_time(g_dec):.1f} us, encoder alone {replay_time(g_enc):.1f} us, two branches of one graph {replay_time(g_both):.1f} us "
      f"(decoder issued first: {replay_time(g_both_r):.1f} us)")
