"""Capture golden vectors from the REAL reference (build container only).

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)

What runs here is the reference's own Python (hash_encoding.py, hash_encoding_wtmk_bit.py,
activation.py, nerf/hidden_models.py, nerf/utils_wtmk_disen.py, nerf/network_wtmk_tcnn.py,
nerf/renderer_wtmk.py), imported from a scratch copy of /root/reference so that nothing is
written into the read-only tree.  Only arrays are saved -- no reference source travels.

Shims (none of them changes arithmetic):
  * torch.tensor(device='cuda') at module scope (hash_encoding.py:8-9, hash_encoding_wtmk_bit.py:9-10)
    is redirected to the CPU while importing;
  * modules that are absent from this image and unused by the captured functions are stubbed
    with MagicMock (trimesh, cv2, imageio, tensorboardX, mcubes, torch_ema, lpips, torchmetrics);
  * torchvision.transforms.Normalize is a 4-line (x-mean)/std stand-in (hidden_models.py:13);
  * G8/G9 only: `tinycudann` and `raymarching` -- the two native dependencies that cannot run here --
    are stand-ins backed by the build's own CPU oracle, so those two vectors pin the reference's
    *glue* (channel placement of the codebook add, message=None branch, background mix, depth
    normalisation, staging), not the arithmetic inside the stand-ins.
"""
import os
import shutil
import sys
import tempfile
import types
from unittest.mock import MagicMock

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import closed_form as cf  # noqa: E402
from oracle import field_ref as fr  # noqa: E402
from oracle import raymarch_ref as rm  # noqa: E402

REF = "/root/reference"


def scratch_reference():
    tmp = tempfile.mkdtemp(prefix="refcopy_")
    for rel in ("hash_encoding.py", "hash_encoding_wtmk_bit.py", "activation.py", "msgencoder.py"):
        shutil.copy(os.path.join(REF, rel), os.path.join(tmp, rel))
    os.makedirs(os.path.join(tmp, "nerf"))
    for rel in ("hidden_models.py", "utils_wtmk_disen.py", "utils_wtmk.py", "provider_wtmk.py", "network_wtmk_tcnn.py", "renderer_wtmk.py"):
        shutil.copy(os.path.join(REF, "nerf", rel), os.path.join(tmp, "nerf", rel))
    open(os.path.join(tmp, "nerf", "__init__.py"), "w").close()
    return tmp


class _CpuTensorCtor:
    """Redirect torch.tensor(..., device='cuda') to the CPU while the reference modules import."""

    def __enter__(self):
        self.orig = torch.tensor

        def ctor(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return self.orig(*a, **k)

        torch.tensor = ctor

    def __exit__(self, *exc):
        torch.tensor = self.orig


def install_stubs():
    for name in ("trimesh", "cv2", "imageio", "tensorboardX", "mcubes", "torch_ema", "lpips", "torchmetrics",
                 "torchmetrics.functional", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, x):
            m = torch.tensor(self.mean, dtype=x.dtype).view(-1, 1, 1)
            s = torch.tensor(self.std, dtype=x.dtype).view(-1, 1, 1)
            return (x - m) / s

    class ToPILImage:
        """torchvision's conversion for float CHW tensors: mul(255).byte(), HWC, mode RGB."""

        def __call__(self, pic):
            from PIL import Image
            return Image.fromarray(pic.detach().cpu().mul(255).byte().permute(1, 2, 0).contiguous().numpy(), mode="RGB")

    tvt.Normalize = Normalize
    tvt.ToPILImage = ToPILImage
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt


def tcnn_standin():
    """`tinycudann` stand-in built on the oracle's fp32 MLP / SH (see module docstring)."""
    mod = types.ModuleType("tinycudann")

    class Network(torch.nn.Module):
        def __init__(self, n_input_dims, n_output_dims, network_config):
            super().__init__()
            w = network_config["n_neurons"]
            self.n_in, self.n_out = n_input_dims, n_output_dims
            self.widths = ((w, 32),) + ((w, w),) * (network_config["n_hidden_layers"] - 1) + ((16, w),)
            self.params = torch.nn.Parameter(torch.zeros(sum(a * b for a, b in self.widths)))

        def forward(self, x):
            if x.shape[1] < 32:
                x = torch.cat([x, torch.ones(x.shape[0], 32 - x.shape[1], dtype=x.dtype)], dim=-1)
            return fr.mlp(x, fr.split_mlp_params(self.params, self.widths))[:, :self.n_out]

    class Encoding(torch.nn.Module):
        def __init__(self, n_input_dims, encoding_config):
            super().__init__()
            self.n_output_dims = encoding_config["degree"] ** 2
            self.params = torch.nn.Parameter(torch.zeros(0))

        def forward(self, x):
            return fr.sh4(x * 2 - 1)

    mod.Network, mod.Encoding = Network, Encoding
    return mod


def raymarching_standin():
    """`raymarching` stand-in: the C oracle behind the reference module's function names."""
    mod = types.ModuleType("raymarching")
    t = torch.from_numpy

    def near_far_from_aabb(o, d, aabb, min_near=0.2):
        n, f = rm.near_far_from_aabb(o.detach().numpy(), d.detach().numpy(), aabb.numpy(), min_near)
        return t(n), t(f)

    def march_rays_train(o, d, bound, bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                         align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        ctr = step_counter.numpy() if step_counter is not None else None
        out = rm.march_rays_train(o.detach().numpy(), d.detach().numpy(), bound, bitfield.numpy(), C, H, nears.numpy(),
                                  fars.numpy(), ctr, mean_count, perturb, align, force_all_rays, dt_gamma, max_steps)
        return tuple(t(np.ascontiguousarray(a)) for a in out)

    def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        return fr._CompositeTrain.apply(sigmas.float(), rgbs.float(), deltas, rays, T_thresh)

    def march_rays(n_alive, n_step, rays_alive, rays_t, o, d, bound, bitfield, C, H, nears, fars, align=-1,
                   perturb=False, dt_gamma=0, max_steps=1024):
        out = rm.march_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), o.numpy(), d.numpy(), bound,
                            bitfield.numpy(), C, H, nears.numpy(), fars.numpy(), align, perturb, dt_gamma, max_steps)
        return tuple(t(a) for a in out)

    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        rm.composite_rays(n_alive, n_step, rays_alive.numpy(), rays_t.numpy(), sigmas.detach().numpy(),
                          rgbs.detach().numpy(), deltas.numpy(), weights_sum.numpy(), depth.numpy(), image.numpy(), T_thresh)

    for f in (near_far_from_aabb, march_rays_train, composite_rays_train, march_rays, composite_rays):
        setattr(mod, f.__name__, f)
    mod.morton3D = lambda c: t(rm.morton3D(c.numpy()))
    mod.morton3D_invert = lambda i: t(rm.morton3D_invert(i.numpy()))
    mod.packbits = lambda g, th, bf=None: t(rm.packbits(g.numpy(), th))
    return mod


ball_scene, orbit_rays = cf.ball_scene, cf.orbit_rays


def grid_maintenance(ref_rend):
    """G11: the reference's own NeRFRenderer.mark_untrained_grid / update_extra_state (renderer_wtmk.py:380-538) on a 32^3, two-cascade
    grid with a closed-form density field and closed-form draws (closed_form.PatchedDraws): -1 marks, two full updates, one partial
    update; grid, bitfield, mean density and mean count after every call."""

    class Field(ref_rend.NeRFRenderer):
        def density(self, x, message=None):
            return cf.grid_density(x, message)

    r = Field(bound=2, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=0.6, bg_radius=-1)
    G = 32
    r.grid_size = G
    r.density_grid = torch.zeros(r.cascade, G ** 3)
    r.density_bitfield = torch.zeros(r.cascade * G ** 3 // 8, dtype=torch.uint8)
    poses, intr = cf.grid_poses()
    out = {"poses": poses, "intrinsics": intr, "grid_size": np.int32(G)}
    r.mark_untrained_grid(poses, intr, S=16)
    out["grid_marked"] = r.density_grid.numpy().copy()
    r.iter_density = 14
    r.local_step = 5
    r.step_counter[:5, 0] = torch.tensor([1000, 1203, 990, 1500, 20], dtype=torch.int32)
    with cf.PatchedDraws() as draws:
        for k in range(3):          # iter_density 14, 15: every cell; 16: a random quarter + as many occupied cells
            if k == 2:
                r.local_step = 20
                r.step_counter[:, 0] = torch.arange(16, dtype=torch.int32) * 100 + 7
            r.update_extra_state(message=None, decay=0.95, S=16)
            out[f"grid_{k}"] = r.density_grid.numpy().copy()
            out[f"bitfield_{k}"] = r.density_bitfield.numpy().copy()
            out[f"mean_density_{k}"] = np.float64(r.mean_density)
            out[f"mean_count_{k}"] = np.int64(r.mean_count)
            out[f"draw_calls_{k}"] = np.int64(draws.calls)
    np.savez_compressed(os.path.join(HERE, "g11_grid_maintenance.npz"), **out)


def main():
    tmp = scratch_reference()
    sys.path.insert(0, tmp)
    install_stubs()
    sys.modules["tinycudann"] = tcnn_standin()
    sys.modules["raymarching"] = raymarching_standin()
    if "--only-g11" in sys.argv:
        with _CpuTensorCtor():
            from nerf import renderer_wtmk as ref_rend
        grid_maintenance(ref_rend)
        shutil.rmtree(tmp)
        print("g11 written to", HERE)
        return
    with _CpuTensorCtor():
        import hash_encoding as ref_he
        import hash_encoding_wtmk_bit as ref_cb
        import activation as ref_act
        from nerf import hidden_models as ref_hm
        from nerf import utils_wtmk_disen as ref_utils
        from nerf import network_wtmk_tcnn as ref_net
    torch.manual_seed(0)
    x = torch.from_numpy(cf.points())

    # ---- G1: base encoder ------------------------------------------------------------------
    enc = ref_he.HashEmbedder(bounding_box=(0, 1), n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                              base_resolution=16, finest_resolution=2048)
    with torch.no_grad():
        for l in range(16):
            enc.embeddings[l].weight.copy_(torch.from_numpy(cf.table(l)))
        feats = enc(x)
        res, rows, w = [], [], []
        for l in range(16):
            r = torch.floor(enc.base_resolution * enc.b ** l)
            vmin, vmax, hashed, _ = ref_he.get_voxel_vertices(x, enc.bounding_box, r, enc.log2_hashmap_size)
            res.append(float(r)); rows.append(hashed.numpy()); w.append(((x - vmin) / (vmax - vmin)).numpy())
    np.savez_compressed(os.path.join(HERE, "g1_base_encoder.npz"), resolutions=np.array(res, np.float32),
                        rows=np.stack(rows).astype(np.int32), weights=np.stack(w), features=feats.numpy())

    # ---- G2: codebook encoder fwd + table grads ----------------------------------------------
    g2 = {}
    for D in (32, 48):
        cb = ref_cb.HashEmbedder(bounding_box=(0, 1), n_levels=D * 2, n_features_per_level=2, log2_hashmap_size=19,
                                 base_resolution=2048, finest_resolution=2048, message_dim=D)
        with torch.no_grad():
            for l in range(2 * D):
                cb.embeddings[l].weight.copy_(torch.from_numpy(cf.table(100 + l, scale=0.05)))
        rvec = torch.from_numpy(np.random.RandomState(5).randn(256, 2).astype(np.float32))
        for k, msg in enumerate(cf.messages(D)):
            cb.zero_grad(set_to_none=True)
            out = cb(x, torch.from_numpy(msg))
            (out * rvec).sum().backward()
            g2[f"out_D{D}_m{k}"] = out.detach().numpy()
            sel = [2 * i + int(msg[i]) for i in range(D)]
            uns = [2 * i + 1 - int(msg[i]) for i in range(D)]
            assert all(cb.embeddings[j].weight.grad is None for j in uns)
            g0 = cb.embeddings[sel[0]].weight.grad
            nz = torch.nonzero(g0.abs().sum(-1)).squeeze(-1)
            g2[f"grad_rows_D{D}_m{k}"] = nz.numpy().astype(np.int32)
            g2[f"grad_vals_D{D}_m{k}"] = g0[nz].numpy()
            # every selected table receives the same gradient (same rows, same weights, same upstream grad)
            g2[f"grad_maxdiff_D{D}_m{k}"] = np.float32(max(float((cb.embeddings[j].weight.grad - g0).abs().max()) for j in sel))
        g2[f"resolution_D{D}"] = np.float32(float(torch.floor(cb.base_resolution * cb.b ** 3)))
    g2["rvec"] = rvec.numpy()
    np.savez_compressed(os.path.join(HERE, "g2_codebook.npz"), **g2)

    # ---- G3: spherical harmonics; G4: trunc_exp -------------------------------------------------
    d = torch.from_numpy(cf.unit_dirs())
    sh = ref_he.SHEncoder(3, 4)(d)
    v = torch.linspace(-20, 20, 161, requires_grad=True)
    y = ref_act.trunc_exp(v)
    y.backward(torch.ones_like(y))
    np.savez_compressed(os.path.join(HERE, "g3_g4_sh_truncexp.npz"), sh=sh.numpy(), te_x=v.detach().numpy(),
                        te_y=y.detach().numpy(), te_grad=v.grad.numpy())

    # ---- G5: decoder + normalisation + losses; G6: meters -----------------------------------------
    torch.manual_seed(3)
    dec = ref_hm.get_hidden_decoder_multi_views(num_bits=1, redundancy=1, num_blocks=8, input_ch=3, channels=64)
    img = torch.rand(32, 12, 12, 3)
    msg = torch.from_numpy(cf.messages(32)[2])
    inp = img.permute(0, 3, 1, 2).clone().requires_grad_(True)
    decoded = dec(ref_hm.normalize_img(inp))
    lossw = torch.nn.functional.binary_cross_entropy_with_logits(decoded * 10.0, msg.unsqueeze(-1), reduction="mean")
    lossw.backward()
    acc = ref_utils.BIT_ACC(device="cpu")
    acc.update(decoded.detach().permute(1, 0), msg[None])
    pm = ref_utils.PSNRMeter()
    pm.update(img, (img * 0.9 + 0.02))
    np.savez_compressed(os.path.join(HERE, "g5_g6_decoder_meters.npz"), img=img.numpy(), msg=msg.numpy(),
                        normalized=ref_hm.normalize_img(inp).detach().numpy(), decoded=decoded.detach().numpy(),
                        lossw=np.float32(lossw.item()), grad_img=inp.grad.numpy(), bit_acc=np.float32(acc.measure()),
                        psnr=np.float32(pm.measure()),
                        **{"dec." + k: v_.numpy() for k, v_ in dec.state_dict().items()})

    # ---- G7: get_rays ---------------------------------------------------------------------------------
    pose, intr, inds = orbit_rays(64)
    torch.manual_seed(0)
    rays = ref_utils.get_rays(torch.from_numpy(pose)[None], intr, 400, 400, -1)
    o_all, d_all = rays["rays_o"], rays["rays_d"]
    np.savez_compressed(os.path.join(HERE, "g7_get_rays.npz"), pose=pose, intrinsics=intr, inds=inds,
                        rays_o=o_all[0, inds].numpy(), rays_d=d_all[0, inds].numpy())

    # ---- G8/G9: glue semantics with stand-ins ---------------------------------------------------------
    D = 32
    torch.manual_seed(0)
    model = ref_net.NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1,
                                message_dim=D, n_views=1)
    grid, bitfield, C = ball_scene()
    with torch.no_grad():
        for l in range(16):
            model.encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(l)))
        for l in range(2 * D):
            model.msg_encoder.embeddings[l].weight.copy_(torch.from_numpy(cf.table(100 + l, scale=0.05)))
        model.sigma_net.params.copy_(torch.from_numpy(cf.mlp_params(3072, 1337)))
        model.color_net.params.copy_(torch.from_numpy(cf.mlp_params(7168, 1338)))
        model.density_grid.copy_(torch.from_numpy(grid))
        model.density_bitfield.copy_(torch.from_numpy(bitfield))
    model.train()
    msg = torch.from_numpy(cf.messages(D)[2])
    pts = torch.from_numpy(cf.points() * 2 - 1).float()
    sig_m, rgb_m = model(pts, d, msg)
    sig_0, rgb_0 = model(pts, d, None)
    o, dd = o_all[:, inds].contiguous(), d_all[:, inds].contiguous()
    for p in model.msg_encoder.parameters():
        p.grad = None
    out = model.render(o, dd, msg, staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0, max_steps=1024,
                       some_unrelated_flag=123)
    gvec = torch.from_numpy(np.random.RandomState(9).randn(1, 64, 3).astype(np.float32))
    (out["image"] * gvec).sum().backward()
    sel0 = model.msg_encoder.embeddings[int(msg[0])].weight.grad
    nz = torch.nonzero(sel0.abs().sum(-1)).squeeze(-1)
    out_staged = model.render(o, dd, msg, staged=True, max_ray_batch=24, bg_color=1, perturb=False, force_all_rays=True,
                              dt_gamma=0, max_steps=1024)
    out_clean = model.render(o, dd, None, staged=False, bg_color=1, perturb=False, force_all_rays=True, dt_gamma=0,
                             max_steps=1024)
    model.eval()
    with torch.no_grad():
        out_eval = model.render(o, dd, msg, staged=False, bg_color=1, perturb=False, dt_gamma=0, max_steps=1024)
    np.savez_compressed(os.path.join(HERE, "g8_g9_glue.npz"), pts=pts.numpy(), dirs=d.numpy(), msg=msg.numpy(),
                        sigma_msg=sig_m.detach().numpy(), rgb_msg=rgb_m.detach().numpy(), sigma_clean=sig_0.detach().numpy(),
                        rgb_clean=rgb_0.detach().numpy(), rays_o=o.numpy(), rays_d=dd.numpy(), gvec=gvec.numpy(),
                        image=out["image"].detach().numpy(), depth=out["depth"].detach().numpy(),
                        weights_sum=out["weights_sum"].detach().numpy(), image_staged=out_staged["image"].detach().numpy(),
                        depth_staged=out_staged["depth"].detach().numpy(), image_clean=out_clean["image"].detach().numpy(),
                        image_eval=out_eval["image"].numpy(), depth_eval=out_eval["depth"].numpy(),
                        cb_grad_rows=nz.numpy().astype(np.int32), cb_grad_vals=sel0[nz].numpy(),
                        state_dict_keys=np.array(sorted(model.state_dict().keys())),
                        state_dict_shapes=np.array([str(tuple(model.state_dict()[k].shape)) for k in sorted(model.state_dict().keys())]))
    # ---- G10: block selection (provider_wtmk.process_image: JPEG compressibility) and rand_poses -----------------------
    sys.modules.setdefault("scipy.spatial.transform", __import__("scipy.spatial.transform", fromlist=["x"]))
    with _CpuTensorCtor():
        from nerf import provider_wtmk as ref_prov
    rng = np.random.RandomState(21)
    yy, xx = np.meshgrid(np.linspace(0, 1, 96), np.linspace(0, 1, 120), indexing="ij")
    img = np.stack([0.5 + 0.4 * np.sin(6 * xx), 0.5 + 0.4 * np.cos(5 * yy), 0.5 + 0.3 * np.sin(9 * xx * yy)], -1)
    noise = rng.rand(96, 120, 3) - 0.5
    for (r, c, amp) in [(1, 2, 0.6), (5, 7, 0.9), (3, 3, 0.3), (6, 0, 0.45), (0, 9, 0.75), (7, 5, 0.2), (2, 8, 0.5)]:
        img[r * 12:(r + 1) * 12, c * 12:(c + 1) * 12] += amp * noise[r * 12:(r + 1) * 12, c * 12:(c + 1) * 12]
    img = torch.from_numpy(np.clip(img, 0, 1).astype(np.float32))[None]
    coords, bh, bw = ref_prov.process_image(img, 8, 10, 6)
    torch.manual_seed(5)
    rp = ref_prov.rand_poses(4, "cpu", radius=2.5)
    np.savez_compressed(os.path.join(HERE, "g10_blocks.npz"), image=img.numpy(), coords=coords.numpy(), bh=np.int32(bh), bw=np.int32(bw), rand_poses=rp.numpy())
    # ---- G11: density-grid maintenance ------------------------------------------------------------------------------------
    from nerf import renderer_wtmk as ref_rend
    grid_maintenance(ref_rend)
    shutil.rmtree(tmp)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
