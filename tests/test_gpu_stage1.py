"""N3 (stage-1 support): gradients of every parameter of the clean model (base tables, both MLPs) against the oracle's
autograd, and a short clean-model training run through the same kernels."""
import numpy as np
import pytest
import torch

import closed_form as cf
from oracle import field_ref as fr

pytestmark = pytest.mark.gpu


def _clean_model(bound=1.0):
    from nerf_signature_amd.stage1 import CleanNeRFNetwork
    m = CleanNeRFNetwork(bound=bound, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)
    grid, bitfield, C = cf.ball_scene(bound=bound)
    with torch.no_grad():
        for l in range(16):
            m.encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(l)))
        m.sigma_net.params.copy_(torch.from_numpy(cf.mlp_params(3072, 1337)))
        m.color_net.params.copy_(torch.from_numpy(cf.mlp_params(7168, 1338)))
        m.density_grid.copy_(torch.from_numpy(grid))
        m.density_bitfield.copy_(torch.from_numpy(bitfield))
    return m.cuda().train(), bitfield, C


def test_all_parameter_gradients_vs_oracle():
    m, bitfield, C = _clean_model()
    trainable = {n for n, p in m.named_parameters() if p.requires_grad and p.numel()}
    assert trainable == {f"encoder.embeddings.{l}.weight" for l in range(16)} | {"sigma_net.params", "color_net.params"}
    rng = np.random.RandomState(0)
    M = 3001
    pts = torch.from_numpy((rng.rand(M, 3) * 2 - 1).astype(np.float32))
    pts[:3] = torch.tensor([[-1., -1, -1], [1, 1, 1], [0.5, -0.25, 0]])
    dirs = torch.from_numpy(cf.unit_dirs(M, seed=9))
    gs = torch.from_numpy(rng.randn(M).astype(np.float32))
    gc = torch.from_numpy(rng.randn(M, 3).astype(np.float32))
    P = {"bound": 1.0, "base_tables": [e.weight.detach().cpu().clone().requires_grad_(True) for e in m.encoder.embeddings],
         "cb_tables": [], "sigma_params": m.sigma_net.params.detach().cpu().clone().requires_grad_(True),
         "color_params": m.color_net.params.detach().cpu().clone().requires_grad_(True)}
    s0, c0 = fr.field_forward(pts, dirs, None, P)
    ((s0 * gs).sum() + (c0 * gc).sum()).backward()
    s1, c1 = m(pts.cuda(), dirs.cuda())
    np.testing.assert_allclose(s1.detach().cpu().numpy(), s0.detach().numpy(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(c1.detach().cpu().numpy(), c0.detach().numpy(), rtol=0, atol=1e-3)
    ((s1 * gs.cuda()).sum() + (c1 * gc.cuda()).sum()).backward()

    def rel(a, b):
        return float((a.cpu() - b).norm() / (b.norm() + 1e-30))
    # MLP weight gradients (sums over all points: ReLU-boundary flips average out)
    assert rel(m.sigma_net.params.grad, P["sigma_params"].grad) < 2e-3
    assert rel(m.color_net.params.grad, P["color_params"].grad) < 2e-3
    assert float(m.color_net.params.grad[6144 + 3 * 64:].abs().max()) == 0.0        # rows 3..15 of the padded colour head: no gradient
    for l in (0, 3, 7, 11, 15):
        g1, g0 = m.encoder.embeddings[l].weight.grad, P["base_tables"][l].grad
        assert torch.equal(g1.cpu() != 0, g0 != 0) or float(((g1.cpu() != 0) != (g0 != 0)).float().mean()) < 1e-4, l
        assert rel(g1, g0) < 5e-3, l


def test_clean_model_trains_through_the_same_kernels():
    """A few optimisation steps on rays of the ball scene: the loss against a fixed target image goes down."""
    from nerf_signature_amd.stage1 import CleanLoop
    m, bitfield, C = _clean_model()
    pose, intr, _ = cf.orbit_rays(1, seed=2)
    rr, cc = np.meshgrid(np.arange(184, 216), np.arange(184, 216), indexing="ij")     # 32x32 pixels through the middle of the ball
    inds = torch.from_numpy((rr * 400 + cc).reshape(-1).astype(np.int64))
    o, d = fr.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, inds[None])
    target = torch.tensor([0.2, 0.5, 0.8]).view(1, 1, 3).expand(1, 1024, 3).contiguous().cuda()
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    loop = CleanLoop(m, opt, dict(dt_gamma=0, max_steps=1024), update_extra_interval=1000)
    loop.global_step = 1                        # keep the synthetic density grid (no refresh at step 0)
    data = {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": target, "perturb": False, "force_all_rays": True}
    losses = [float(loop.step(data)[1].detach()) for _ in range(12)]
    assert losses[-1] < 0.5 * losses[0], losses
    m.update_extra_state()                      # the density-grid refresh of the loop runs on the trained field
    assert m.iter_density == 1
