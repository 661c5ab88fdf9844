"""Seeded synthetic point clouds (no dataset exists on either box).

CAD-like clouds: area-uniform samples on a random mixture of 2-6 primitives
(sphere, box faces, cylinder, plane patch), rescaled like the reference's
ModelNet40 sampler does (sample_modelnet.py:47-48: p - min(p); p / max(p)).
"""
import numpy as np


def _sphere(rng, n):
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
    return v * rng.uniform(0.2, 0.6) + rng.uniform(-0.4, 0.4, size=3)


def _box(rng, n):
    ext = rng.uniform(0.2, 0.8, size=3)
    areas = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]] * 2)
    face = rng.choice(6, size=n, p=areas / areas.sum())
    p = rng.uniform(-0.5, 0.5, size=(n, 3))
    ax = face % 3
    p[np.arange(n), ax] = np.where(face < 3, -0.5, 0.5)
    return p * ext + rng.uniform(-0.3, 0.3, size=3)


def _cylinder(rng, n):
    r, h = rng.uniform(0.1, 0.4), rng.uniform(0.3, 1.0)
    t = rng.uniform(0, 2 * np.pi, size=n)
    p = np.stack([r * np.cos(t), r * np.sin(t), rng.uniform(-h / 2, h / 2, size=n)], axis=1)
    perm = rng.permutation(3)
    return p[:, perm] + rng.uniform(-0.3, 0.3, size=3)


def _plane(rng, n):
    a, b = rng.uniform(0.3, 1.0, size=2)
    p = np.stack([rng.uniform(-a / 2, a / 2, size=n), rng.uniform(-b / 2, b / 2, size=n), np.zeros(n)], axis=1)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return p @ q.T + rng.uniform(-0.3, 0.3, size=3)


_PRIMS = (_sphere, _box, _cylinder, _plane)


def cad_cloud(seed, n=8192):
    """One (n,3) float32 cloud in [0,1]^3, deterministic in ``seed``."""
    rng = np.random.default_rng(seed)
    m = int(rng.integers(2, 7))
    w = rng.uniform(0.5, 1.5, size=m)
    counts = np.floor(w / w.sum() * n).astype(int)
    counts[0] += n - counts.sum()
    parts = [_PRIMS[int(rng.integers(0, 4))](rng, int(c)) for c in counts]
    p = np.concatenate(parts, axis=0)
    p = p[rng.permutation(n)]
    p = p - p.min()
    p = p / p.max()
    return p.astype(np.float32)


def cad_batch(first_seed, count, n=8192):
    """(count,n,3) float32; cloud i uses seed first_seed+i (default_rng(11+i))."""
    return np.stack([cad_cloud(first_seed + i, n) for i in range(count)])


def room_cloud(seed, n=None):
    """Room-like cloud for configs[3] (S3DIS Area_1 stand-in): floor, ceiling, four walls and a few boxes
    (furniture), area-uniform samples, metres; ``n`` points (default: seeded, 0.5-1 M).  (n,3) float32, deterministic."""
    rng = np.random.default_rng(seed)
    if n is None:
        n = int(rng.integers(500_000, 1_000_001))
    W, D, H = rng.uniform(4, 12), rng.uniform(4, 10), rng.uniform(2.6, 3.6)
    rects = [((0, 0, 0), (W, 0, 0), (0, D, 0)), ((0, 0, H), (W, 0, 0), (0, D, 0)),          # floor, ceiling
             ((0, 0, 0), (W, 0, 0), (0, 0, H)), ((0, D, 0), (W, 0, 0), (0, 0, H)),          # walls
             ((0, 0, 0), (0, D, 0), (0, 0, H)), ((W, 0, 0), (0, D, 0), (0, 0, H))]
    for _ in range(int(rng.integers(4, 10))):                                                # boxes standing on the floor
        e = rng.uniform(0.3, 1.8, size=3) * np.array([1, 1, 0.7])
        o = np.array([rng.uniform(0, W - e[0]), rng.uniform(0, D - e[1]), 0.0])
        ex, ey, ez = np.diag(e)
        rects += [(o, ex, ey), (o + ez, ex, ey), (o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez)]
    areas = np.array([np.linalg.norm(np.cross(np.asarray(u, float), np.asarray(v, float))) for _, u, v in rects])
    counts = np.floor(areas / areas.sum() * n).astype(int)
    counts[0] += n - counts.sum()
    parts = []
    for (o, u, v), c in zip(rects, counts):
        ab = rng.uniform(0, 1, size=(int(c), 2))
        parts.append(np.asarray(o, float) + ab[:, :1] * np.asarray(u, float) + ab[:, 1:] * np.asarray(v, float))
    p = np.concatenate(parts) + rng.normal(0, 0.002, size=(n, 3))                            # sensor noise, 2 mm
    return p[rng.permutation(n)].astype(np.float32)
