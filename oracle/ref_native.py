"""oracle/ref_native.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Loader for oracle/_ref/_raymarching_ref.so: the REFERENCE's own native extension (its raymarching.cu + bindings.cpp,
built for gfx950 by oracle/build_ref.py), i.e. `kind: "reference"` in the sense of the measurement contract.  The
functions below only restate the buffer allocation of the reference's Python wrappers
(/root/reference/raymarching/raymarching.py:19-49, 196-233, 329-373) around the module's ten entry points
(/root/reference/raymarching/src/bindings.cpp:5-18); every number they return is computed by the reference's kernels.
torch CUDA tensors in and out.  The reference launches on the legacy default stream, so callers stay on torch's
default stream.  Only tests/ may import this module.
"""
import importlib.machinery
import importlib.util
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_ref", "_raymarching_ref.so")
_mod = None


def available():
    return os.path.exists(SO)


def module():
    global _mod
    if _mod is None:
        loader = importlib.machinery.ExtensionFileLoader("_raymarching_ref", SO)
        spec = importlib.util.spec_from_loader("_raymarching_ref", loader)
        _mod = importlib.util.module_from_spec(spec)
        loader.exec_module(_mod)
    return _mod


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    N = rays_o.shape[0]
    nears, fars = torch.empty(N, device=rays_o.device), torch.empty(N, device=rays_o.device)
    module().near_far_from_aabb(rays_o.contiguous(), rays_d.contiguous(), aabb, N, float(min_near), nears, fars)
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    N = rays_o.shape[0]
    coords = torch.empty(N, 2, device=rays_o.device)
    module().sph_from_ray(rays_o.contiguous(), rays_d.contiguous(), float(radius), N, coords)
    return coords


def morton3D(coords):
    N = coords.shape[0]
    out = torch.empty(N, dtype=torch.int32, device=coords.device)
    module().morton3D(coords.int().contiguous(), N, out)
    return out


def morton3D_invert(indices):
    N = indices.shape[0]
    out = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    module().morton3D_invert(indices.int().contiguous(), N, out)
    return out


def packbits(grid, thresh):
    grid = grid.contiguous()
    C, H3 = grid.shape
    N = C * H3 // 8
    out = torch.empty(N, dtype=torch.uint8, device=grid.device)
    module().packbits(grid, N, float(thresh), out)
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, noises=None, dt_gamma=0.0, max_steps=1024, M=None, counter=None):
    """Full-capacity call (M = N * max_steps unless given).  Returns xyzs, dirs, deltas [M,.] (zero beyond the used range), rays [N,3] in
    the kernel's atomic arrival order, counter [2]."""
    N, dev = rays_o.shape[0], rays_o.device
    M = N * max_steps if M is None else M
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    rays = torch.full((N, 3), -7, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev) if counter is None else counter
    noises = torch.zeros(N, device=dev) if noises is None else noises
    module().march_rays_train(rays_o.contiguous(), rays_d.contiguous(), bitfield, float(bound), float(dt_gamma), int(max_steps), N, C, H, M, nears, fars,
                              xyzs, dirs, deltas, rays, counter, noises)
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    M, N, dev = sigmas.shape[0], rays.shape[0], sigmas.device
    ws, depth, image = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
    module().composite_rays_train_forward(sigmas.contiguous(), rgbs.contiguous(), deltas, rays, M, N, float(T_thresh), ws, depth, image)
    return ws, depth, image


def composite_rays_train_backward(g_ws, g_image, sigmas, rgbs, deltas, rays, ws, image, T_thresh=1e-4):
    M, N = sigmas.shape[0], rays.shape[0]
    g_sigmas, g_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
    module().composite_rays_train_backward(g_ws.contiguous(), g_image.contiguous(), sigmas.contiguous(), rgbs.contiguous(), deltas, rays, ws, image, M, N,
                                           float(T_thresh), g_sigmas, g_rgbs)
    return g_sigmas, g_rgbs


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars, align=-1, noises=None, dt_gamma=0.0, max_steps=1024):
    dev = rays_o.device
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    noises = torch.zeros(n_alive, device=dev) if noises is None else noises
    module().march_rays(n_alive, n_step, rays_alive, rays_t, rays_o.contiguous(), rays_d.contiguous(), float(bound), float(dt_gamma), int(max_steps), C, H,
                        bitfield, nears, fars, xyzs, dirs, deltas, noises)
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    module().composite_rays(n_alive, n_step, float(T_thresh), rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(), deltas, weights_sum, depth, image)
