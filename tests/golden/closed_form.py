"""Closed-form, seed-free generators shared by tests/golden/make_golden.py and the tests.

Hash tables are far too large to commit (4 MiB each), so both the script that captured the
golden outputs from the reference and the tests that replay them rebuild the same tables
from this integer formula.
"""
import numpy as np

from oracle import raymarch_ref as rm

T = 1 << 19


def table(level, scale=0.5, T=T):
    """[T,2] fp32 table for `level`: a multiplicative integer hash of (row, feature, level) mapped to [-scale, scale)."""
    idx = (np.arange(T * 2, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(level) * np.uint64(0x9E3779B9) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    idx = (idx ^ (idx >> np.uint64(15))) * np.uint64(2246822519) & np.uint64(0xFFFFFFFF)
    idx = idx ^ (idx >> np.uint64(13))
    return ((idx.astype(np.float64) / 4294967296.0 - 0.5) * 2 * scale).astype(np.float32).reshape(T, 2)


def points(n=256, seed=7):
    """[n,3] fp32 points in [0,1] with the edge cases the encoders are sensitive to:
    exact 0 and 1, and exact / nearly exact cell boundaries k/res of several levels."""
    rng = np.random.RandomState(seed)
    x = rng.rand(n, 3).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2] = (0.0, 1.0, 0.5)
    x[3] = np.float32(1.0) - np.float32(2 ** -24)
    for r, res in enumerate((16, 22, 153, 561, 2047, 2048)):
        k = np.array([1, res // 2, res - 1], dtype=np.float32)
        x[4 + r] = k / np.float32(res)
        x[10 + r] = np.nextafter(k / np.float32(res), np.float32(0), dtype=np.float32)
        x[16 + r] = np.nextafter(k / np.float32(res), np.float32(2), dtype=np.float32)
    return np.clip(x, 0, 1)


def unit_dirs(n=256, seed=11):
    rng = np.random.RandomState(seed)
    d = rng.randn(n, 3).astype(np.float32)
    d[0] = (1, 0, 0)
    d[1] = (0, -1, 0)
    d[2] = (0, 0, 1)
    return (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)


def messages(D):
    rng = np.random.RandomState(100 + D)
    return np.stack([np.zeros(D), np.ones(D), rng.randint(0, 2, D)]).astype(np.float32)


def mlp_params(n, seed):
    """Flat fp32 parameter vector, uniform in +-sqrt(6/(64+32)) (Xavier-like), seeded."""
    rng = np.random.RandomState(seed)
    a = np.sqrt(6.0 / 96.0)
    return rng.uniform(-a, a, n).astype(np.float32)


def ball_scene(bound=1.0, radius=0.5, H=128):
    """Scene S0 of SURVEY.md 8(d): density 100 inside a ball, packed at thresh min(mean, 10)."""
    C = 1 + int(np.ceil(np.log2(bound)))
    ii = np.arange(H, dtype=np.int32)
    gx, gy, gz = np.meshgrid(ii, ii, ii, indexing="ij")
    coords = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], -1)
    idx = rm.morton3D(coords)
    grid = np.zeros((C, H ** 3), np.float32)
    for c in range(C):
        b = min(2 ** c, bound)
        p = (2 * (coords.astype(np.float32) + 0.5) / H - 1) * b
        grid[c, idx] = np.where(np.linalg.norm(p, axis=-1) < radius, 100.0, 0.0)
    thresh = min(float(grid.clip(min=0).mean()), 10.0)
    return grid, rm.packbits(grid, thresh), C


def orbit_rays(n, radius=3.2248, seed=0, H=400, W=400, focal=555.56):
    """n rays of one orbit camera looking at the origin (pixel indices drawn with a seeded RandomState)."""
    rng = np.random.RandomState(seed)
    th, ph = 1.1, 0.7
    c = np.array([radius * np.sin(th) * np.sin(ph), radius * np.cos(th), radius * np.sin(th) * np.cos(ph)], np.float32)
    fwd = -c / np.linalg.norm(c)
    right = np.cross(fwd, np.array([0, 1, 0], np.float32)); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, -up, fwd, c
    inds = rng.randint(0, H * W, size=n)
    return pose, np.array([focal, focal, W / 2, H / 2], np.float32), inds.astype(np.int64)


def uniform(n, stream):
    """n fp32 values in [0,1) from an integer hash of (index, stream): the stand-in for torch.rand_like / torch.randint draws
    wherever the reference draws on its own device generator (density-grid jitter, renderer_wtmk.py:478,514; partial-update cell
    choice :488,492) -- the golden capture and the tests patch both functions with this generator, so the draws are identical on
    the CPU, on the GPU and in the reference run."""
    idx = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(stream) * np.uint64(0x85EBCA6B) + np.uint64(777)) & np.uint64(0xFFFFFFFF)
    idx = (idx ^ (idx >> np.uint64(16))) * np.uint64(2246822519) & np.uint64(0xFFFFFFFF)
    idx = (idx ^ (idx >> np.uint64(13))) * np.uint64(3266489917) & np.uint64(0xFFFFFFFF)
    idx = idx ^ (idx >> np.uint64(16))
    return ((idx >> np.uint64(8)).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)


class PatchedDraws:
    """Context manager: torch.rand_like / torch.randint answer from `uniform` (one stream per call, in call order)."""

    def __enter__(self):
        import torch
        self.torch, self.calls = torch, 0
        self.orig = (torch.rand_like, torch.randint)

        def rand_like(t, **kw):
            self.calls += 1
            return torch.from_numpy(uniform(t.numel(), self.calls)).view(t.shape).to(device=t.device, dtype=t.dtype)

        def randint(low, high, size, dtype=torch.int64, device=None, **kw):
            self.calls += 1
            n = int(np.prod(size))
            v = low + np.floor(uniform(n, self.calls).astype(np.float64) * (high - low)).astype(np.int64)
            return torch.from_numpy(v).view(*size).to(device=device, dtype=dtype)

        torch.rand_like, torch.randint = rand_like, randint
        return self

    def __exit__(self, *exc):
        self.torch.rand_like, self.torch.randint = self.orig


def grid_density(x, message=None):
    """Closed-form density field for the density-grid maintenance vectors: only +, -, *, clamp on single elements, so the CPU, the
    GPU and the reference run produce identical fp32 values.  x: torch [n,3]."""
    r2 = x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1] + x[:, 2] * x[:, 2]
    return {"sigma": (30.0 * (1.0 - r2 * 1.5)).clamp(min=0.0) + 0.004 * (x[:, 0] + 2.0).clamp(min=0.0)}


def grid_poses():
    """Three camera-to-world poses around the origin (radius 1.3, looking inwards) + intrinsics (fx, fy, cx, cy) with a narrow view."""
    poses = []
    for th, ph in ((1.1, 0.7), (1.4, 2.9), (0.6, 4.4)):
        c = np.array([1.3 * np.sin(th) * np.sin(ph), 1.3 * np.cos(th), 1.3 * np.sin(th) * np.cos(ph)], np.float32)
        fwd = -c / np.linalg.norm(c)
        right = np.cross(fwd, np.array([0, 1, 0], np.float32)); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, -up, fwd, c
        poses.append(pose)
    return np.stack(poses), np.array([300.0, 300.0, 100.0, 100.0], np.float32)
