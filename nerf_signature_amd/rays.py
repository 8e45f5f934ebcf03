"""get_rays on the device: the step just before the render path.

Mirror of get_rays in /root/reference/nerf/utils_wtmk_disen.py:59-143 (same arguments, same result dict: 'rays_o',
'rays_d' [B,N,3], 'inds' [B,N], and 'inds_coarse' with an error map).  Index selection keeps the reference's torch
calls (randint / multinomial / patch offsets, now on the device the poses live on); the ray arithmetic -- the
reference's [B, H*W] meshgrid, gathers and ~15 elementwise ops -- is one kernel (rg_get_rays)."""
import torch

from . import _native as nv


@torch.no_grad()
def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1):
    device = poses.device
    if not poses.is_cuda:
        raise ValueError("get_rays: poses must be on the GPU")
    B = poses.shape[0]
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    results = {}
    if N > 0:
        N = min(N, H * W)
        if patch_size > 1:      # utils_wtmk_disen.py:85-102
            num_patch = N // (patch_size ** 2)
            inds_x = torch.randint(0, H - patch_size, size=[num_patch], device=device)
            inds_y = torch.randint(0, W - patch_size, size=[num_patch], device=device)
            inds = torch.stack([inds_x, inds_y], dim=-1)
            pi, pj = torch.meshgrid(torch.arange(patch_size, device=device), torch.arange(patch_size, device=device), indexing="ij")
            offsets = torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1)
            inds = (inds.unsqueeze(1) + offsets.unsqueeze(0)).view(-1, 2)
            inds = inds[:, 0] * W + inds[:, 1]
            inds = inds.expand([B, inds.shape[0]])
        elif error_map is None:  # :104-106
            inds = torch.randint(0, H * W, size=[N], device=device).expand([B, N])
        else:                    # :108-119
            inds_coarse = torch.multinomial(error_map.to(device), N, replacement=False)
            inds_x, inds_y = inds_coarse // 128, inds_coarse % 128
            sx, sy = H / 128, W / 128
            inds_x = (inds_x * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=H - 1)
            inds_y = (inds_y * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=W - 1)
            inds = inds_x * W + inds_y
            results["inds_coarse"] = inds_coarse
        inds = inds.contiguous().long()
        n = inds.shape[1]
        ind_ptr = nv.ptr(inds)
    else:
        n = H * W
        inds = torch.arange(H * W, device=device).expand([B, H * W])
        ind_ptr = None
    results["inds"] = inds
    P = poses.contiguous().float()
    rays_o = torch.empty(B, n, 3, dtype=torch.float32, device=device)
    rays_d = torch.empty(B, n, 3, dtype=torch.float32, device=device)
    nv.call("rg_get_rays", nv.ptr(P), fx, fy, cx, cy, int(H), int(W), ind_ptr, B, n, nv.ptr(rays_o), nv.ptr(rays_d), nv.stream())
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    return results


class DeviceRaySampler:
    """The loader step of the training loop on the device (rg_sample_rays): a store of poses [P,4,4] and (optionally) their images
    [P,H*W,3] resident in HBM; every call writes one batch -- pose (step * stride + offset) mod P, N uniformly drawn pixels, their rays
    and ground-truth colours -- into caller-owned buffers, with `step` read from a device counter.  Nothing in the call depends on a
    host value, so trainer.GraphedWatermarkLoop captures it at the head of its step (`content_sampler=`): the per-step hand-over of
    rays costs one 5 us launch inside the graph instead of a randint, a ray kernel, a gather and three copies between two replays."""

    def __init__(self, poses, images, intrinsics, H, W, n_rays, stride=1, offset=0, seed=0):
        if not poses.is_cuda:
            raise ValueError("DeviceRaySampler: poses must be on the GPU")
        self.poses = poses.contiguous().float()
        self.images = None if images is None else images.contiguous().float().view(self.poses.shape[0], H * W, 3)
        self.intr = tuple(float(v) for v in intrinsics)
        self.H, self.W, self.n_rays, self.stride, self.offset, self.seed = int(H), int(W), int(n_rays), int(stride), int(offset), int(seed)

    @torch.no_grad()
    def sample_into(self, step_counter, rays_o, rays_d, gt=None, inds_out=None, pose_out=None):
        """step_counter: int32 device tensor [1] (or None = step 0).  rays_o / rays_d / gt: float32 buffers of n_rays * 3 elements."""
        for t in (rays_o, rays_d, gt):
            if t is not None and (t.numel() != self.n_rays * 3 or t.dtype != torch.float32):
                raise ValueError(f"DeviceRaySampler: buffers must hold {self.n_rays} x 3 float32 values")
        if gt is not None and self.images is None:
            raise ValueError("DeviceRaySampler: no image store to take the ground truth from")
        fx, fy, cx, cy = self.intr
        nv.call("rg_sample_rays", nv.ptr(self.poses), self.poses.shape[0], nv.ptr(self.images), fx, fy, cx, cy, self.H, self.W, self.n_rays,
                nv.ptr(step_counter), self.stride, self.offset, self.seed, nv.ptr(rays_o), nv.ptr(rays_d), nv.ptr(gt), nv.ptr(inds_out), nv.ptr(pose_out),
                nv.stream())
