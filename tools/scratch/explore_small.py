"""GPU-side exploration: how many steps does the small (oracle-sized) configuration need to get off chance?"""
import os, sys, copy
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import closed_form as cf
import test_gpu_render as T
from nerf_signature_amd import trainer
from nerf_signature_amd.trainer import BIT_ACC

for block, lw, lr in ((4, 0.005, 1e-2), (6, 0.005, 1e-2), (8, 0.005, 1e-2), (4, 0.05, 1e-2), (6, 0.05, 1e-2), (4, 0.005, 3e-3), (6, 0.005, 3e-3), (4, 1.0, 1e-2)):
  for seed in (0, 1, 2):
    steps, ncont, scale = 200, 64, 1e-4
    torch.manual_seed(seed)
    m, bitfield, C = T._model()
    with torch.no_grad():
        for l in range(64):
            m.msg_encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(100 + l, scale=scale)))
    bo, bd, co, cd, _ = T._data(n_content=ncont, block=block)
    kw = dict(dt_gamma=0.0, max_steps=1024)
    with torch.no_grad():
        gt = m.render(co.cuda(), cd.cuda(), None, staged=False, bg_color=1, perturb=False, force_all_rays=True, **kw)["image"].clamp(0, 1)
    data = {"watermark": {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}, "content": {"rays_o": co.cuda(), "rays_d": cd.cuda(), "images": gt}}
    rng = np.random.RandomState(1234)
    msgs = [torch.from_numpy(rng.randint(0, 2, 32).astype(np.float32)) for _ in range(steps)]
    opt = torch.optim.Adam(m.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / steps, 1))
    loop = trainer.WatermarkLoop(m, opt, kw, lambda_w=lw, lambda_i=1.0, lr_scheduler=sched)
    tr = []
    for k, msg in enumerate(msgs):
        out = loop.step(data, msg)
        if k % 25 == 0 or k == steps - 1:
            tr.append(round(float(out[4].detach()), 3))
    acc = BIT_ACC()
    with torch.no_grad():
        for s in range(40):
            msg = torch.from_numpy(np.random.RandomState(9000 + s).randint(0, 2, 32).astype(np.float32)).cuda()
            _, _, _, dec, _, _, _ = trainer.eval_step(m, data["watermark"], msg, kw, render_whole=False)
            acc.update(dec.permute(1, 0), msg[None])
        img = m.render(co.cuda(), cd.cuda(), msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, **kw)["image"]
    print(f"block {block} lambda_w {lw} lr {lr} seed {seed}: bit acc {acc.measure():.4f}  psnr {-10*np.log10(float(((img-gt)**2).mean())):.2f}  lossw {tr}", flush=True)
