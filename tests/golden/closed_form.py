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
