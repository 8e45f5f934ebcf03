"""Are two runs of the stage-1 loops bit-identical?  (diagnostic: python tools/stage1_determinism.py)  Each loop twice from the same initial state, 5 steps, the
parameters compared bit for bit; then eager vs captured, tensor by tensor."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch

import test_gpu_stage1 as T
from nerf_signature_amd.stage1 import CleanLoop, GraphedCleanLoop

o, d = T._patch_rays(16)
target = torch.tensor([0.2, 0.5, 0.8]).view(1, 3).expand(256, 3).contiguous().cuda()
data = {"rays_o": o.cuda()[None], "rays_d": d.cuda()[None], "images": target[None], "perturb": False, "force_all_rays": True}


def run(kind):
    m, _, _ = T._clean_model()
    if kind == "eager":
        loop = CleanLoop(m, T._adam(m), T.KW, update_extra_interval=10 ** 9)
        loop.global_step = 1
        losses = [float(loop.step(data)[1].detach()) for _ in range(5)]
    else:
        loop = GraphedCleanLoop(m, T._adam(m), T.KW, n_rays=256, update_extra_interval=0, perturb=False, capture=(kind == "captured"))
        losses = [float(loop.step(data)) for _ in range(5)]
    torch.cuda.synchronize()
    return losses, [p.detach().clone() for p in m.trainable()]


res = {}
for kind in ("eager", "captured", "explicit-uncaptured"):
    a, b = run(kind), run(kind)
    same = [torch.equal(x, y) for x, y in zip(a[1], b[1])]
    worst = max(float((x - y).abs().max()) for x, y in zip(a[1], b[1]))
    print(f"{kind}: losses equal {a[0] == b[0]}; tensors bit-identical {sum(same)}/18; max |diff| {worst:.3e}; losses {a[0]}")
    res[kind] = a
for k in ("captured", "explicit-uncaptured"):
    diffs = [float((x - y).abs().max()) for x, y in zip(res["eager"][1], res[k][1])]
    frac = [float(((x - y).abs() > 2e-5).float().mean()) for x, y in zip(res["eager"][1], res[k][1])]
    print(f"eager vs {k}: max |diff| per tensor " + " ".join(f"{v:.1e}" for v in diffs))
    print(f"   fraction > 2e-5 per tensor " + " ".join(f"{v:.1e}" for v in frac))
