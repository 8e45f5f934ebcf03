"""Host-side mirror of the reference's `raymarching` module on top of libnerfsig's C ABI.

Same ten callables, same argument meaning and return values as
/root/reference/raymarching/raymarching.py (cited per class), so `import raymarching` call sites in
nerf/renderer_wtmk.py work unchanged.  Differences that are part of the contract:
  * `rays` from march_rays_train is in ray-id order with prefix-sum offsets (the reference's order is
    whatever its atomics produced, raymarching.cu:405-406);
  * no N*max_steps zero-filled staging buffers and no torch.cuda.empty_cache(): the point count is read
    back once (one 8-byte D2H, the reference's `.item()`, raymarching.py:224) and the outputs are
    allocated at their final padded size;
  * kernels run on torch's current stream; arguments are validated (device, dtype, contiguity).
"""
import os

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _native as nv

_fwd32 = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


def _dev(t):
    return t if t.is_cuda else t.cuda()


def _f32c(t):
    return t.contiguous().float() if t.dtype != torch.float32 or not t.is_contiguous() else t


# ----------------------------------------------------------------------------------------- utils

class _near_far_from_aabb(Function):
    """raymarching.py:19-47 -> rm_near_far_from_aabb."""

    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o = _dev(rays_o).contiguous().view(-1, 3)
        rays_d = _dev(rays_d).contiguous().view(-1, 3)
        aabb = _f32c(_dev(aabb))
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        nv.call("rm_near_far_from_aabb", nv.ptr(rays_o), nv.ptr(rays_d), nv.ptr(aabb), N, float(min_near), nv.ptr(nears),
                nv.ptr(fars), nv.stream())
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


def near_far_into(rays_o, rays_d, aabb, min_near, nears, fars):
    """near_far_from_aabb written into existing [N] float32 tensors (a step marched ahead re-uses its buffers)."""
    N = rays_o.shape[0]
    nv.call("rm_near_far_from_aabb", nv.ptr(rays_o), nv.ptr(rays_d), nv.ptr(_f32c(aabb)), N, float(min_near), nv.ptr(nears), nv.ptr(fars), nv.stream())
    return nears, fars


class _sph_from_ray(Function):
    """raymarching.py:52-78 -> rm_sph_from_ray."""

    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, radius):
        rays_o = _dev(rays_o).contiguous().view(-1, 3)
        rays_d = _dev(rays_d).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        nv.call("rm_sph_from_ray", nv.ptr(rays_o), nv.ptr(rays_d), float(radius), N, nv.ptr(coords), nv.stream())
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    """raymarching.py:83-102 -> rm_morton3D."""

    @staticmethod
    def forward(ctx, coords):
        coords = _dev(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        nv.call("rm_morton3D", nv.ptr(coords), N, nv.ptr(indices), nv.stream())
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    """raymarching.py:106-124 -> rm_morton3D_invert."""

    @staticmethod
    def forward(ctx, indices):
        indices = _dev(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        nv.call("rm_morton3D_invert", nv.ptr(indices), N, nv.ptr(coords), nv.stream())
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    """raymarching.py:129-153 -> rm_packbits."""

    @staticmethod
    @_fwd32
    def forward(ctx, grid, thresh, bitfield=None):
        grid = _dev(grid).contiguous()
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        else:
            ctx.mark_dirty(bitfield)      # written in place through its raw pointer: bump its version (keys of kept samples compare it)
        nv.check(bitfield, torch.uint8, "bitfield")
        nv.call("rm_packbits", nv.ptr(grid), N, float(thresh), nv.ptr(bitfield), nv.stream())
        return bitfield


packbits = _packbits.apply


# ----------------------------------------------------------------------------------------- training

_SCAN_WRITE_MAX = None


def scan_write_max_rays():
    """Largest ray count rm_march_train_scan_write takes (its N + 1 offsets live in LDS)."""
    global _SCAN_WRITE_MAX
    if _SCAN_WRITE_MAX is None:      # NERFSIG_MARCH_FUSED=0 | nf: the stand-alone scan + write launches (A/B measurements)
        _SCAN_WRITE_MAX = 0 if os.environ.get("NERFSIG_MARCH_FUSED", "1") in ("0", "nf") else int(nv.fn("rm_march_train_scan_write_max_rays")())
    return _SCAN_WRITE_MAX


def fused_limits():
    """True: the training march computes near / far itself (rm_march_train_count_nf).  NERFSIG_MARCH_FUSED=0 | sw: a launch of its own."""
    return os.environ.get("NERFSIG_MARCH_FUSED", "1") not in ("0", "sw")


def march_rays_train_device(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, counter, noises, dt_gamma,
                            max_steps, capacity=None, out=None, limits=None):
    """The enqueues of the training march with no host synchronisation.

    Returns (counts, t_rec, rays, write) where `write(M)` fills xyzs/dirs/deltas of M rows -- freshly allocated, or the
    tensors of `out` = (xyzs, dirs, deltas, rays) when given.  `counter` (int32[2]) receives (total points, N) on the device.
    capacity: the row count is known up front (no host read of the total in between), so for ray counts whose offsets fit in LDS the
    prefix sum rides in the write launch (rm_march_train_scan_write) instead of a single-workgroup launch of its own.
    limits = (aabb, min_near): `nears` / `fars` are OUTPUTS, filled by the walk itself (rm_march_train_count_nf) instead of a
    near_far_from_aabb launch in front of it."""
    N = rays_o.shape[0]
    dev = rays_o.device
    counts = torch.empty(N, dtype=torch.int32, device=dev)
    t_rec = torch.empty(N * max_steps, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev) if out is None else out[3]
    s = nv.stream()
    if limits is not None:
        nv.call("rm_march_train_count_nf", nv.ptr(rays_o), nv.ptr(rays_d), nv.ptr(_f32c(limits[0])), float(limits[1]), nv.ptr(density_bitfield), float(bound),
                float(dt_gamma), int(max_steps), N, int(C), int(H), nv.ptr(noises), nv.ptr(nears), nv.ptr(fars), nv.ptr(counts), nv.ptr(t_rec), s)
    else:
        nv.call("rm_march_train_count", nv.ptr(rays_o), nv.ptr(rays_d), nv.ptr(density_bitfield), float(bound), float(dt_gamma),
                int(max_steps), N, int(C), int(H), nv.ptr(nears), nv.ptr(fars), nv.ptr(noises), nv.ptr(counts), nv.ptr(t_rec), s)
    fused = capacity is not None and 1 <= N <= scan_write_max_rays()
    if not fused and N > 16384:      # many rays (a staged full-image render): the two-launch scan of 4096-ray workgroups
        sums = torch.empty(int(nv.fn("rm_march_train_scan_blocks")(N)), dtype=torch.int32, device=dev)
        nv.call("rm_march_train_scan_wide", nv.ptr(counts), N, nv.ptr(rays), nv.ptr(counter), nv.ptr(sums), s)
    elif not fused:
        nv.call("rm_march_train_scan", nv.ptr(counts), N, nv.ptr(rays), nv.ptr(counter), s)

    def write(M):
        if out is not None:
            xyzs, dirs, deltas = out[:3]
            if xyzs.shape[0] != M or rays.shape[0] != N:
                raise ValueError(f"march buffers hold {xyzs.shape[0]} points / {rays.shape[0]} rays, the march needs {M} / {N}")
        else:
            xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
            dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
            deltas = torch.empty(M, 2, dtype=torch.float32, device=dev)
        if fused:
            nv.call("rm_march_train_scan_write", nv.ptr(rays_o), nv.ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), M,
                    nv.ptr(nears), nv.ptr(noises), nv.ptr(t_rec), nv.ptr(counts), nv.ptr(rays), nv.ptr(counter), nv.ptr(xyzs), nv.ptr(dirs),
                    nv.ptr(deltas), nv.stream())
        else:
            nv.call("rm_march_train_write", nv.ptr(rays_o), nv.ptr(rays_d), float(bound), float(dt_gamma), int(max_steps), N, int(C),
                    int(H), M, nv.ptr(nears), nv.ptr(noises), nv.ptr(t_rec), nv.ptr(rays), nv.ptr(counter), nv.ptr(xyzs),
                    nv.ptr(dirs), nv.ptr(deltas), nv.stream())
        return xyzs, dirs, deltas

    return counts, t_rec, rays, write


def march_rays_train_capacity(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter, capacity, perturb=False,
                              dt_gamma=0, max_steps=1024, out=None, limits=None):
    """march_rays_train with force_all_rays semantics but NO host synchronisation: the point buffers have `capacity` rows
    (a caller-chosen bound on the padded point count); the real total lands in step_counter[0] on the device, rows past
    it are zero, and a ray that does not fit is dropped exactly like the reference's bounded mode (raymarching.cu:416) --
    the caller checks step_counter[0] <= capacity after the fact.  This is what makes a training step graph-capturable."""
    rays_o = rays_o.contiguous().view(-1, 3).float()
    rays_d = rays_d.contiguous().view(-1, 3).float()
    N = rays_o.shape[0]
    noises = torch.rand(N, dtype=torch.float32, device=rays_o.device) if perturb else None
    _, _, rays, write = march_rays_train_device(rays_o, rays_d, bound, density_bitfield, C, H, _f32c(nears), _f32c(fars), step_counter, noises,
                                                dt_gamma, max_steps, capacity=int(capacity), out=out, limits=limits)
    xyzs, dirs, deltas = write(int(capacity))
    return xyzs, dirs, deltas, rays


def padded_point_count(m, align=128):
    """The reference's `m += align - m % align` (raymarching.py:225-226)."""
    return m + align - m % align


class _march_rays_train(Function):
    """raymarching.py:161-233 -> rm_march_train_count / _scan / _write."""

    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o = _dev(rays_o).contiguous().view(-1, 3)
        rays_d = _dev(rays_d).contiguous().view(-1, 3)
        density_bitfield = _dev(density_bitfield).contiguous()
        nv.check(density_bitfield, torch.uint8, "density_bitfield")
        if density_bitfield.numel() * 8 != C * H ** 3:
            raise ValueError(f"density_bitfield has {density_bitfield.numel()} bytes, expected C*H^3/8 = {C * H ** 3 // 8}")
        N = rays_o.shape[0]
        if nears.shape[0] != N or fars.shape[0] != N:
            raise ValueError("nears/fars must have one entry per ray")

        M = N * max_steps
        bounded = not force_all_rays and mean_count > 0
        if bounded:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=rays_o.device)
        noises = torch.rand(N, dtype=rays_o.dtype, device=rays_o.device) if perturb else None

        _, _, rays, write = march_rays_train_device(rays_o, rays_d, bound, density_bitfield, C, H, _f32c(nears), _f32c(fars),
                                                    step_counter, noises, dt_gamma, max_steps)
        if not bounded:
            m = int(step_counter[0].item())  # the reference's single D2H read (raymarching.py:224)
            if align > 0:
                m += align - m % align
            M = m
        xyzs, dirs, deltas = write(M)
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    """raymarching.py:238-288 -> rm_composite_train_fwd / _bwd."""

    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas = sigmas.contiguous()
        rgbs = rgbs.contiguous()
        deltas = deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        if rgbs.shape[0] != M or deltas.shape[0] != M:
            raise ValueError("sigmas, rgbs and deltas must have the same number of rows")
        nv.check(rays, torch.int32, "rays", 3)
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        nv.call("rm_composite_train_fwd", nv.ptr(sigmas), nv.ptr(rgbs), nv.ptr(deltas), nv.ptr(rays), M, N, float(T_thresh),
                nv.ptr(weights_sum), nv.ptr(depth), nv.ptr(image), nv.stream())
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        ctx.set_materialize_grads(False)   # the depth gradient is never used: do not make autograd fill a zero tensor for it
        return weights_sum, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is not propagated (raymarching.py:275)
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        grad_weights_sum = torch.zeros_like(weights_sum) if grad_weights_sum is None else grad_weights_sum.contiguous()
        grad_image = torch.zeros_like(image) if grad_image is None else grad_image.contiguous()
        M, N, T_thresh = ctx.dims
        grad_sigmas = torch.empty_like(sigmas)
        grad_rgbs = torch.empty_like(rgbs)
        nv.call("rm_composite_train_bwd", nv.ptr(grad_weights_sum), nv.ptr(grad_image), nv.ptr(sigmas), nv.ptr(rgbs),
                nv.ptr(deltas), nv.ptr(rays), nv.ptr(weights_sum), nv.ptr(image), M, N, float(T_thresh), nv.ptr(grad_sigmas),
                nv.ptr(grad_rgbs), nv.stream())
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


# ----------------------------------------------------------------------------------------- inference

class _march_rays(Function):
    """raymarching.py:297-344 -> rm_march."""

    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
                perturb=False, dt_gamma=0, max_steps=1024):
        rays_o = _dev(rays_o).contiguous().view(-1, 3)
        rays_d = _dev(rays_d).contiguous().view(-1, 3)
        nv.check(rays_alive, torch.int32, "rays_alive")
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        xyzs = torch.empty(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        dirs = torch.empty(M, 3, dtype=rays_o.dtype, device=rays_o.device)
        deltas = torch.empty(M, 2, dtype=rays_o.dtype, device=rays_o.device)
        noises = torch.rand(n_alive, dtype=rays_o.dtype, device=rays_o.device) if perturb else None
        nv.call("rm_march", int(n_alive), int(n_step), nv.ptr(rays_alive), nv.ptr(rays_t), nv.ptr(rays_o), nv.ptr(rays_d), float(bound),
                float(dt_gamma), int(max_steps), int(C), int(H), nv.ptr(density_bitfield), nv.ptr(near), nv.ptr(far), nv.ptr(xyzs),
                nv.ptr(dirs), nv.ptr(deltas), nv.ptr(noises), M, nv.stream())
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    """raymarching.py:349-368 -> rm_composite (weights_sum, depth, image, rays_alive, rays_t updated in place)."""

    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        nv.call("rm_composite", int(n_alive), int(n_step), float(T_thresh), nv.ptr(rays_alive), nv.ptr(rays_t),
                nv.ptr(sigmas.contiguous()), nv.ptr(rgbs.contiguous()), nv.ptr(deltas), nv.ptr(weights_sum), nv.ptr(depth),
                nv.ptr(image), nv.stream())
        return tuple()


composite_rays = _composite_rays.apply


def compact_alive(rays_alive):
    """Device-side replacement of `rays_alive[rays_alive >= 0]` (nerf/renderer_wtmk.py:363).
    Returns (compacted int32 tensor of the same capacity, device int32[1] survivor count)."""
    out = torch.empty_like(rays_alive)
    n_out = torch.empty(1, dtype=torch.int32, device=rays_alive.device)
    nv.call("rm_compact_alive", nv.ptr(rays_alive), rays_alive.shape[0], nv.ptr(out), nv.ptr(n_out), nv.stream())
    return out, n_out
