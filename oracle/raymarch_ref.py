"""oracle/raymarch_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of oracle/raymarch_ref.c (the scalar CPU restatement of
/root/reference/raymarching/src/raymarching.cu) with the allocation and padding
rules of the reference's Python wrappers (/root/reference/raymarching/raymarching.py).
Inputs/outputs are numpy arrays.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_raymarch.so")

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)
_b = ctypes.POINTER(ctypes.c_uint8)
_u32 = ctypes.c_uint32
_fl = ctypes.c_float


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "raymarch_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.oracle_near_far_from_aabb.argtypes = [_f, _f, _f, _u32, _fl, _f, _f]
        L.oracle_sph_from_ray.argtypes = [_f, _f, _fl, _u32, _f]
        L.oracle_morton3D.argtypes = [_i, _u32, _i]
        L.oracle_morton3D_invert.argtypes = [_i, _u32, _i]
        L.oracle_packbits.argtypes = [_f, _u32, _fl, _b]
        L.oracle_march_rays_train.argtypes = [_f, _f, _b, _fl, _fl, _u32, _u32, _u32, _u32, _u32, _f, _f, _f, _f, _f,
                                              _i, _i, _f]
        L.oracle_composite_rays_train_forward.argtypes = [_f, _f, _f, _i, _u32, _u32, _fl, _f, _f, _f]
        L.oracle_composite_rays_train_backward.argtypes = [_f, _f, _f, _f, _f, _i, _f, _f, _u32, _u32, _fl, _f, _f]
        L.oracle_march_rays.argtypes = [_u32, _u32, _i, _f, _f, _f, _fl, _fl, _u32, _u32, _u32, _b, _f, _f, _f, _f,
                                        _f, _f]
        L.oracle_composite_rays.argtypes = [_u32, _u32, _fl, _i, _f, _f, _f, _f, _f, _f, _f]
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching.py:19-49."""
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().oracle_near_far_from_aabb(_p(o, _f), _p(d, _f), _p(_f32(aabb), _f), N, min_near, _p(nears, _f), _p(fars, _f))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching.py:52-80."""
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().oracle_sph_from_ray(_p(o, _f), _p(d, _f), radius, N, _p(coords, _f))
    return coords


def morton3D(coords):
    """raymarching.py:83-104."""
    c = np.ascontiguousarray(coords, dtype=np.int32)
    N = c.shape[0]
    out = np.empty(N, np.int32)
    lib().oracle_morton3D(_p(c, _i), N, _p(out, _i))
    return out


def morton3D_invert(indices):
    """raymarching.py:106-126."""
    ind = np.ascontiguousarray(indices, dtype=np.int32)
    N = ind.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().oracle_morton3D_invert(_p(ind, _i), N, _p(out, _i))
    return out


def packbits(grid, thresh, bitfield=None):
    """raymarching.py:129-155."""
    g = _f32(grid)
    N = g.size // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    lib().oracle_packbits(_p(g, _f), N, thresh, _p(bitfield, _b))
    return bitfield


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0.0, max_steps=1024, noises=None):
    """raymarching.py:161-235 (allocation, M bound, `align` padding rule included).

    `noises` may be given explicitly so that tests can exercise the perturbed path
    with a fixed noise vector (the reference draws torch.rand)."""
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    N = o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if step_counter is None:
        step_counter = np.zeros(2, np.int32)
    if noises is None:
        noises = np.random.rand(N).astype(np.float32) if perturb else np.zeros(N, np.float32)
    noises = _f32(noises)
    lib().oracle_march_rays_train(_p(o, _f), _p(d, _f), _p(grid, _b), bound, dt_gamma, max_steps, N, C, H, M,
                                  _p(_f32(nears), _f), _p(_f32(fars), _f), _p(xyzs, _f), _p(dirs, _f), _p(deltas, _f),
                                  _p(rays, _i), _p(step_counter, _i), _p(noises, _f))
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def march_counts(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, dt_gamma=0.0, max_steps=1024,
                 noises=None):
    """Count pass only (no point buffers): per-ray sample counts in ray-id order and their total."""
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    N = o.shape[0]
    rays = np.empty((N, 3), np.int32)
    counter = np.zeros(2, np.int32)
    noises = np.zeros(N, np.float32) if noises is None else _f32(noises)
    lib().oracle_march_rays_train(_p(o, _f), _p(d, _f), _p(grid, _b), bound, dt_gamma, max_steps, N, C, H, 0,
                                  _p(_f32(nears), _f), _p(_f32(fars), _f), None, None, None, _p(rays, _i),
                                  _p(counter, _i), _p(noises, _f))
    return rays, int(counter[0])


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """raymarching.py:238-266."""
    s, c, dl = _f32(sigmas), _f32(rgbs), _f32(deltas)
    r = np.ascontiguousarray(rays, dtype=np.int32)
    M, N = s.shape[0], r.shape[0]
    ws, depth, image = np.empty(N, np.float32), np.empty(N, np.float32), np.empty((N, 3), np.float32)
    lib().oracle_composite_rays_train_forward(_p(s, _f), _p(c, _f), _p(dl, _f), _p(r, _i), M, N, T_thresh, _p(ws, _f),
                                              _p(depth, _f), _p(image, _f))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh=1e-4):
    """raymarching.py:268-288 (grad_depth is ignored by the reference, :275)."""
    s, c, dl = _f32(sigmas), _f32(rgbs), _f32(deltas)
    r = np.ascontiguousarray(rays, dtype=np.int32)
    M, N = s.shape[0], r.shape[0]
    gs, gc = np.zeros_like(s), np.zeros_like(c)
    lib().oracle_composite_rays_train_backward(_p(_f32(grad_ws), _f), _p(_f32(grad_image), _f), _p(s, _f), _p(c, _f),
                                               _p(dl, _f), _p(r, _i), _p(_f32(weights_sum), _f), _p(_f32(image), _f),
                                               M, N, T_thresh, _p(gs, _f), _p(gc, _f))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars,
               align=-1, perturb=False, dt_gamma=0.0, max_steps=1024, noises=None):
    """raymarching.py:297-346."""
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    if noises is None:
        noises = np.random.rand(n_alive).astype(np.float32) if perturb else np.zeros(n_alive, np.float32)
    ra = np.ascontiguousarray(rays_alive, dtype=np.int32)
    lib().oracle_march_rays(n_alive, n_step, _p(ra, _i), _p(_f32(rays_t), _f), _p(o, _f), _p(d, _f), bound, dt_gamma,
                            max_steps, C, H, _p(grid, _b), _p(_f32(nears), _f), _p(_f32(fars), _f), _p(xyzs, _f),
                            _p(dirs, _f), _p(deltas, _f), _p(_f32(noises), _f))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching.py:349-370; rays_alive, rays_t, weights_sum, depth, image are updated in place
    (they must be C-contiguous arrays of the right dtype)."""
    for a, t in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32),
                 (image, np.float32)):
        assert a.dtype == t and a.flags.c_contiguous
    lib().oracle_composite_rays(n_alive, n_step, T_thresh, _p(rays_alive, _i), _p(rays_t, _f), _p(_f32(sigmas), _f),
                                _p(_f32(rgbs), _f), _p(_f32(deltas), _f), _p(weights_sum, _f), _p(depth, _f),
                                _p(image, _f))
