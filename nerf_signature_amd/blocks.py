"""Watermark-block selection and the clean-render pre-pass: the data preparation on either side of the
render path in the reference's NeRFDataset_Disen (/root/reference/nerf/provider_wtmk.py).

  * rand_poses            -- orbit poses (provider_wtmk.py:60-96)
  * compression_ratio     -- per-block JPEG size ratio, PIL on the host (provider_wtmk.py:146-171)
  * process_image         -- split an image into num_rows x num_cols blocks, keep the `num_selections` blocks that JPEG
                             compresses worst (lowest ratio), return their pixel rectangles (provider_wtmk.py:173-218)
  * block_rays            -- the rays of those rectangles, [D, bh, bw, 3] (provider_wtmk.py:481-494)
  * clean_render          -- the "ground truth" of the watermark stage: the same model rendered without a message,
                             staged in max_ray_batch chunks (provider_wtmk.py:408-416,455-460), rays generated on the device
"""
from io import BytesIO

import numpy as np
import torch


def rand_poses(size, device, radius=1.0, theta_range=(np.pi / 3, 2 * np.pi / 3), phi_range=(0, 2 * np.pi)):
    """Random orbit cameras looking at the origin, [size,4,4] camera-to-world (provider_wtmk.py:61-96 -- same two uniform
    draws, polar angle first).  The reference builds the look-at frame with two cross products against the world's -y axis;
    on an orbit that frame has a closed form in the two angles, written out here: with s/c = sin/cos,
        centre  = r (s_t s_p,  c_t, s_t c_p)       forward = -centre / r
        right   = (-c_p, 0, s_p)                    up      = (c_t s_p, -s_t, c_t c_p)
    (valid for polar angles strictly inside (0, pi), where the reference's frame is defined at all; golden G10)."""
    lo_t, hi_t = theta_range
    lo_p, hi_p = phi_range
    theta = torch.rand(size, device=device) * (hi_t - lo_t) + lo_t
    phi = torch.rand(size, device=device) * (hi_p - lo_p) + lo_p
    st, ct, sp, cp = torch.sin(theta), torch.cos(theta), torch.sin(phi), torch.cos(phi)
    zero = torch.zeros_like(theta)
    poses = torch.zeros(size, 4, 4, dtype=torch.float32, device=device)
    poses[:, :3, 0] = torch.stack([-cp, zero, sp], dim=-1)
    poses[:, :3, 1] = torch.stack([ct * sp, -st, ct * cp], dim=-1)
    poses[:, :3, 2] = torch.stack([-st * sp, -ct, -st * cp], dim=-1)
    poses[:, :3, 3] = torch.stack([radius * st * sp, radius * ct, radius * st * cp], dim=-1)
    poses[:, 3, 3] = 1.0
    return poses


def _to_pil(chw):
    """torchvision.transforms.ToPILImage for a float [3,h,w] tensor in [0,1]: mul(255).byte(), HWC, mode RGB."""
    from PIL import Image
    arr = chw.detach().cpu().mul(255).byte().permute(1, 2, 0).contiguous().numpy()
    return Image.fromarray(arr, mode="RGB")


def compression_ratio(blocks):
    """blocks: [1, rows, cols, 3, bh, bw] -> [1, rows, cols, 1]: bytes(JPEG default) / bytes(JPEG optimize, quality 75)."""
    n, rows, cols = blocks.shape[:3]
    flat = blocks.reshape(n * rows * cols, *blocks.shape[3:])
    ratios = []
    for b in flat:
        img = _to_pil(b)
        raw, opt = BytesIO(), BytesIO()
        img.save(raw, format="JPEG")
        img.save(opt, format="JPEG", optimize=True, quality=75)
        ratios.append(float(raw.tell()) / opt.tell())
    return torch.from_numpy(np.stack(ratios, axis=0)).to(blocks.device).view(n, rows, cols, 1)


def process_image(image, num_rows, num_cols, num_selections):
    """image [1,H,W,3] in [0,1] -> (block_coordinates [D,4] = (row0, col0, row1, col1), block_height, block_width)."""
    _, height, width, _ = image.shape
    bh, bw = height // num_rows, width // num_cols
    image = image[:, :bh * num_rows, :bw * num_cols]
    blocks = image.unfold(1, bh, bh).unfold(2, bw, bw)          # [1, rows, cols, 3, bh, bw]
    score = compression_ratio(blocks)
    selected = torch.argsort(score.view(-1))[:num_selections]   # the least compressible blocks first
    rows = torch.div(selected, num_cols, rounding_mode="floor")
    cols = selected % num_cols
    coords = torch.stack((rows * bh, cols * bw, (rows + 1) * bh, (cols + 1) * bw), dim=1)
    return coords, bh, bw


def block_rays(rays_o, rays_d, block_coordinates):
    """rays_o/rays_d [1,H,W,3], block_coordinates [D,4] -> [D,bh,bw,3] each."""
    bo, bd = [], []
    for r0, c0, r1, c1 in block_coordinates.tolist():
        bo.append(rays_o[:, r0:r1, c0:c1, :])
        bd.append(rays_d[:, r0:r1, c0:c1, :])
    return torch.cat(bo, dim=0).contiguous(), torch.cat(bd, dim=0).contiguous()


@torch.no_grad()
def clean_render(model, poses, intrinsics, H, W, render_kwargs, max_ray_batch=4096):
    """[B,H,W,3] clean images of `poses` (message=None), one staged render per pose."""
    from .rays import get_rays
    out = []
    for b in range(poses.shape[0]):
        rays = get_rays(poses[b:b + 1], intrinsics, H, W, -1)
        img = model.render(rays["rays_o"], rays["rays_d"], None, staged=True, max_ray_batch=max_ray_batch, bg_color=None, perturb=False,
                           force_all_rays=True, **render_kwargs)["image"]
        out.append(img.reshape(1, H, W, 3))
    return torch.cat(out, dim=0)
