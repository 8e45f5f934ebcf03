#!/usr/bin/env python
"""Which tiny-cuda-nn parameter layout does a REAL checkpoint use?  (the MLP arithmetic is the one part of the path
nothing in the reference tree can pin -- tinycudann is a third-party CUDA dependency, absent and unversioned.)

The kernels (mlp_pack_weights / field_fwd, csrc/field.hip) assume what INTEGRATION.md section 3 documents: `sigma_net.params` and
`color_net.params` are consecutive [out, in] row-major matrices (network_wtmk_tcnn.py:52-88), the colour net's 31 real inputs are
[16 SH | 15 geometry features] padded to 32 with a constant 1.0, and the SH encoding sees d mapped back from [0,1] to [-1,1] in (x, y, z)
order.  If one of these is wrong a real clean.pth loads key-for-key and renders garbage.  This tool turns "unverified" into a
five-minute check: on ANY machine that has tinycudann, dump a few hundred (x, d) -> (sigma, rgb) pairs of the reference's own model
(--print-dump-script writes the snippet); here -- no GPU, no tinycudann, plain torch on the CPU -- every combination of the layout
hypotheses is evaluated against the dump and the one that reproduces it is named.

    python tools/check_tcnn_layout.py --print-dump-script > dump_field.py       # run THAT next to the reference + tinycudann
    python tools/check_tcnn_layout.py --checkpoint clean.pth --dump field_dump.npz [--convert fixed.pth]
    python tools/check_tcnn_layout.py --self-test                              # synthetic: every hypothesis is recovered from its own dump

--convert writes a checkpoint whose MLP vectors are re-laid-out into the form this repository expects, when the matching hypothesis is
a pure re-layout (transposed matrices, a zero pad column); SH argument conventions cannot be fixed by re-layout and are only reported.
Tolerance: tinycudann's FullyFusedMLP computes in fp16 (weights, activations, accumulation); the fp32 evaluation here agrees with it to
about 1e-2 relative on sigma and 5e-3 absolute on rgb.  The winning hypothesis is far below, the wrong ones are O(1) off."""
import argparse
import itertools
import math
import sys

import numpy as np
import torch

T = 1 << 19
PRIMES = (1, 2654435761, 805459861)
SIGMA_WIDTHS = ((64, 32), (16, 64))
COLOR_WIDTHS = ((64, 32), (64, 64), (16, 64))

DUMP_SCRIPT = r'''# Run on a machine that has tinycudann, next to the reference checkout (python dump_field.py clean.pth [message bits, e.g. 0110...]).
# Writes field_dump.npz: 512 points x, unit directions d, and the reference model's own sigma / rgb for them.
import sys, numpy as np, torch
from nerf.network_wtmk_tcnn import NeRFNetwork
ckpt = torch.load(sys.argv[1], map_location="cpu")
sd = ckpt["model"] if "model" in ckpt else ckpt
D = sum(k.startswith("msg_encoder.embeddings.") for k in sd) // 2 or 16
bound = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
model = NeRFNetwork(bound=bound, cuda_ray=False, message_dim=D, n_views=1).cuda()
print(model.load_state_dict(sd, strict=False))
g = torch.Generator().manual_seed(0)
x = (torch.rand(512, 3, generator=g) * 2 - 1) * bound * 0.9
d = torch.nn.functional.normalize(torch.randn(512, 3, generator=g), dim=-1)
msg = None
if len(sys.argv) > 2 and sys.argv[2] not in ("", "-"):
    msg = torch.tensor([float(c) for c in sys.argv[2]]).cuda()
with torch.no_grad():
    sigma, rgb = model(x.cuda(), d.cuda(), msg)
np.savez("field_dump.npz", x=x.numpy(), d=d.numpy(), sigma=sigma.float().cpu().numpy(), rgb=rgb.float().cpu().numpy(), bound=bound,
         message=np.zeros(0, np.float32) if msg is None else msg.cpu().numpy())
print("wrote field_dump.npz")
'''


# ------------------------------------------------------------------------------------------------ the field, hypothesis by hypothesis

def level_resolutions():
    b = torch.exp((torch.log(torch.tensor(2048.0)) - torch.log(torch.tensor(16.0))) / 15)       # hash_encoding.py:58-60, fp32
    return [int(torch.floor(torch.tensor(16.0) * b ** l)) for l in range(16)]


def encode_level(x01, table, res):
    """hash_encoding.py:24-46,75-111: cell = floor(x / fp32(1/res)), every corner hashed, trilinear weights from the cell's own corners."""
    gs = torch.tensor(1.0 / res, dtype=torch.float32)
    cell = torch.floor(x01 / gs).to(torch.int64)
    lo = cell.to(torch.float32) * gs
    w = (x01 - lo) / ((lo + gs) - lo)
    out = 0.0
    for c in range(8):
        off = torch.tensor([(c >> 2) & 1, (c >> 1) & 1, c & 1])
        v = cell + off
        h = ((v[:, 0] * PRIMES[0]) ^ (v[:, 1] * PRIMES[1]) ^ (v[:, 2] * PRIMES[2])) & (T - 1)
        wc = torch.where(off[0] == 1, w[:, 0], 1 - w[:, 0]) * torch.where(off[1] == 1, w[:, 1], 1 - w[:, 1]) * torch.where(off[2] == 1, w[:, 2], 1 - w[:, 2])
        out = out + table[h] * wc[:, None]
    return out


def features(x, bound, base_tables, cb_tables, message):
    x01 = (x + bound) / (2 * bound)
    feat = torch.cat([encode_level(x01, t, r) for t, r in zip(base_tables, level_resolutions())], dim=-1)
    if message is not None and len(message):
        add = sum(encode_level(x01, cb_tables[2 * i + int(message[i])], 2048) for i in range(len(message)))
        feat = torch.cat([feat[:, :-2], feat[:, -2:] + add], dim=-1)          # network_wtmk_tcnn.py:106
    return feat


def sh4(v):
    x, y, z = v.unbind(-1)
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    return torch.stack([torch.full_like(x, 0.28209479177387814), -0.4886025119029199 * y, 0.4886025119029199 * z, -0.4886025119029199 * x,
                        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.31539156525252005 * (2.0 * zz - xx - yy), -1.0925484305920792 * xz,
                        0.5462742152960396 * (xx - yy), -0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z,
                        -0.4570457994644658 * y * (4 * zz - xx - yy), 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
                        -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy), -0.5900435899266435 * x * (xx - 3 * yy)], dim=-1)


def split(params, widths, layout):
    mats, off = [], 0
    for fo, fi in widths:
        chunk = params[off:off + fo * fi]
        mats.append(chunk.view(fo, fi) if layout == "out_in" else chunk.view(fi, fo).t())
        off += fo * fi
    if off != params.numel():
        raise SystemExit(f"parameter vector has {params.numel()} elements, the architecture needs {off}")
    return mats


def mlp(h, mats):
    for k, W in enumerate(mats):
        h = h @ W.t()
        if k + 1 < len(mats):
            h = torch.relu(h)
    return h


HYPOTHESES = {
    "layout": ("out_in", "in_out"),                     # each matrix [out,in] row-major (assumed) / [in,out] row-major
    "pad": (1.0, 0.0),                                  # value of the colour net's padded 32nd input (assumed 1.0)
    "sh_input": ("minus1_1", "zero_1"),                 # SH evaluated on d (the [0,1] input mapped back, assumed) / on (d+1)/2 as handed over
    "sh_axes": tuple(itertools.permutations(range(3))),  # which component of d is the formula's x, y, z (assumed (0,1,2))
}
ASSUMED = {"layout": "out_in", "pad": 1.0, "sh_input": "minus1_1", "sh_axes": (0, 1, 2)}


def evaluate(feat, d, sigma_params, color_params, hyp):
    h = mlp(feat, split(sigma_params, SIGMA_WIDTHS, hyp["layout"]))
    sigma = torch.exp(h[:, 0])
    v = d if hyp["sh_input"] == "minus1_1" else (d + 1) / 2
    v = v[:, list(hyp["sh_axes"])]
    cin = torch.cat([sh4(v), h[:, 1:], torch.full_like(h[:, :1], hyp["pad"])], dim=-1)
    rgb = torch.sigmoid(mlp(cin, split(color_params, COLOR_WIDTHS, hyp["layout"]))[:, :3])
    return sigma, rgb


def score(sigma, rgb, ref_sigma, ref_rgb):
    es = float(((sigma - ref_sigma).abs() / (ref_sigma.abs() + 1e-3)).median())       # median relative error of sigma
    ec = float((rgb - ref_rgb).abs().max())
    return es, ec


def all_hypotheses():
    keys = list(HYPOTHESES)
    for combo in itertools.product(*(HYPOTHESES[k] for k in keys)):
        yield dict(zip(keys, combo))


def rank(feat, d, sp, cp, ref_sigma, ref_rgb):
    rows = []
    for hyp in all_hypotheses():
        s, c = evaluate(feat, d, sp, cp, hyp)
        es, ec = score(s, c, ref_sigma, ref_rgb)
        rows.append((es + ec, es, ec, hyp))
    rows.sort(key=lambda r: r[0])
    return rows


def describe(hyp):
    diff = [f"{k} = {hyp[k]} (assumed {ASSUMED[k]})" for k in hyp if hyp[k] != ASSUMED[k]]
    return "the documented layout (INTEGRATION.md section 3)" if not diff else "; ".join(diff)


def convert_params(sp, cp, hyp):
    """Re-lay the two vectors out into the documented form, when the hypothesis is a pure re-layout."""
    if hyp["sh_input"] != ASSUMED["sh_input"] or tuple(hyp["sh_axes"]) != ASSUMED["sh_axes"]:
        return None
    out = []
    for params, widths in ((sp, SIGMA_WIDTHS), (cp, COLOR_WIDTHS)):
        mats = [m.clone() for m in split(params, widths, hyp["layout"])]
        if widths is COLOR_WIDTHS and hyp["pad"] == 0.0:
            mats[0][:, 31] = 0.0          # a pad of 0 contributes nothing: the same as a zero weight column under the documented pad of 1
        out.append(torch.cat([m.contiguous().reshape(-1) for m in mats]))
    return out


# ------------------------------------------------------------------------------------------------ entry points

def load_tables(sd, prefix, n):
    return [sd[f"{prefix}.embeddings.{i}.weight"].float() for i in range(n)]


def check(checkpoint, dump, convert=None, quiet=False):
    ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False) if isinstance(checkpoint, str) else checkpoint
    sd = ckpt["model"] if "model" in ckpt else ckpt
    z = np.load(dump) if isinstance(dump, str) else dump
    x, d = torch.from_numpy(np.asarray(z["x"], np.float32)), torch.from_numpy(np.asarray(z["d"], np.float32))
    ref_sigma, ref_rgb = torch.from_numpy(np.asarray(z["sigma"], np.float32)).reshape(-1), torch.from_numpy(np.asarray(z["rgb"], np.float32)).reshape(-1, 3)
    bound = float(z["bound"]) if "bound" in z else 1.0
    message = np.asarray(z["message"]).reshape(-1) if "message" in z else np.zeros(0)
    n_cb = sum(k.startswith("msg_encoder.embeddings.") for k in sd)
    if len(message) and n_cb < 2 * len(message):
        raise SystemExit(f"the dump used a {len(message)}-bit message but the checkpoint holds {n_cb} codebook tables")
    feat = features(x, bound, load_tables(sd, "encoder", 16), load_tables(sd, "msg_encoder", n_cb) if len(message) else [], message)
    sp, cp = sd["sigma_net.params"].float().reshape(-1), sd["color_net.params"].float().reshape(-1)
    rows = rank(feat, d, sp, cp, ref_sigma, ref_rgb)
    best = rows[0]
    ok = best[1] < 3e-2 and best[2] < 2e-2
    unique = len(rows) < 2 or rows[1][0] > 4 * max(best[0], 1e-3)
    if not quiet:
        print(f"{len(rows)} layout hypotheses against {x.shape[0]} dumped points (median relative error of sigma | max absolute error of rgb):")
        for tot, es, ec, hyp in rows[:6]:
            print(f"  {es:9.2e} | {ec:9.2e}   {describe(hyp)}")
        print(f"  ... worst: {rows[-1][1]:.2e} | {rows[-1][2]:.2e}")
        if ok and best[3] == ASSUMED:
            print("RESULT: the checkpoint uses the documented layout -- it loads and renders as it is.")
        elif ok:
            print(f"RESULT: the checkpoint matches ANOTHER layout: {describe(best[3])}.")
            print("        adapt mlp_pack_weights (csrc/field.hip) / oracle.field_ref.split_mlp_params accordingly, or re-lay the checkpoint out with --convert"
                  if convert_params(sp, cp, best[3]) is not None else
                  "        this is a convention of the SH encoding, not a re-layout: change the SH argument handling in csrc/field.hip (k_field_fwd) and oracle/field_ref.py:color")
        else:
            print("RESULT: NO hypothesis reproduces the dump (best errors above the fp16 tolerance): the dump and the checkpoint do not belong together, or the "
                  "layout differs in a way this tool does not enumerate (hidden width, biases, another activation).")
        if ok and not unique:
            print("        (note: a second hypothesis fits almost as well -- e.g. a pad of 0 and of 1 coincide when the checkpoint's pad column is all zero)")
    if convert and ok:
        new = convert_params(sp, cp, best[3])
        if new is None:
            raise SystemExit("--convert: the matching hypothesis is not a pure re-layout")
        sd = dict(sd)
        sd["sigma_net.params"], sd["color_net.params"] = new
        torch.save({**ckpt, "model": sd} if "model" in ckpt else sd, convert)
        if not quiet:
            print(f"wrote {convert} in the documented layout")
    return ok, best[3], rows


def self_test():
    """Every hypothesis is recovered from a dump generated under it (small random tables / weights), and --convert turns a transposed,
    zero-padded checkpoint into one that matches under the documented layout."""
    g = torch.Generator().manual_seed(0)
    sd = {f"encoder.embeddings.{i}.weight": (torch.rand(T, 2, generator=g) - 0.5) for i in range(16)}
    sd.update({f"msg_encoder.embeddings.{i}.weight": (torch.rand(T, 2, generator=g) - 0.5) * 0.1 for i in range(8)})
    for name, widths in (("sigma_net.params", SIGMA_WIDTHS), ("color_net.params", COLOR_WIDTHS)):
        sd[name] = torch.cat([(torch.rand(fo * fi, generator=g) * 2 - 1) * math.sqrt(6.0 / (fo + fi)) for fo, fi in widths])
    x = (torch.rand(256, 3, generator=g) * 2 - 1) * 0.9
    d = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1)
    msg = np.array([1, 0, 0, 1], np.float32)
    feat = features(x, 1.0, load_tables(sd, "encoder", 16), load_tables(sd, "msg_encoder", 8), msg)
    n = 0
    for hyp in all_hypotheses():
        s, c = evaluate(feat, d, sd["sigma_net.params"], sd["color_net.params"], hyp)
        s16, c16 = s.half().float(), c.half().float()           # a dump comes from an fp16 network
        dump = {"x": x.numpy(), "d": d.numpy(), "sigma": s16.numpy(), "rgb": c16.numpy(), "bound": 1.0, "message": msg}
        ok, best, rows = check(sd, dump, quiet=True)
        assert ok and best == hyp, (hyp, best, rows[0][:3])
        n += 1
    hyp = {"layout": "in_out", "pad": 0.0, "sh_input": "minus1_1", "sh_axes": (0, 1, 2)}
    s, c = evaluate(feat, d, sd["sigma_net.params"], sd["color_net.params"], hyp)
    new = convert_params(sd["sigma_net.params"], sd["color_net.params"], hyp)
    s2, c2 = evaluate(feat, d, new[0], new[1], ASSUMED)
    assert torch.allclose(s, s2, rtol=1e-5, atol=1e-6) and torch.allclose(c, c2, rtol=0, atol=1e-6)
    print(f"self-test ok: {n} hypotheses each recovered from their own dump; a transposed, zero-padded checkpoint converts to the documented layout")
    return n


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--checkpoint")
    ap.add_argument("--dump")
    ap.add_argument("--convert")
    ap.add_argument("--print-dump-script", action="store_true")
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.print_dump_script:
        sys.stdout.write(DUMP_SCRIPT)
        return 0
    if a.self_test:
        self_test()
        return 0
    if not (a.checkpoint and a.dump):
        ap.error("--checkpoint and --dump are both needed (or --print-dump-script / --self-test)")
    ok, _, _ = check(a.checkpoint, a.dump, a.convert)
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
