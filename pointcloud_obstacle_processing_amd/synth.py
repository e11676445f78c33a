"""Seeded synthetic clouds for the BASELINE.json configs.

All clouds are float32 (n,3), shuffled so input order carries no spatial
locality.  The field box of C1/C2 is the one in the reference's
minibot_cr18/params.yaml:2-7 (x 0..4.5, y 0..3.78, z -0.5..0.25).
"""
import numpy as np

FIELD_BOX = ((0.0, 4.5), (0.0, 3.78), (-0.5, 0.25))

# name -> (n, leaf, tol, min_cluster, max_cluster)
CONFIGS = {
    "C1": dict(n=10_000, leaf=0.05, tol=0.1, min_cluster=5, max_cluster=20000, kind="uniform_field"),
    "C2": dict(n=100_000, leaf=0.05, tol=0.2, min_cluster=5, max_cluster=20000, kind="uniform_field"),
    "C3": dict(n=1_000_000, leaf=0.02, tol=0.1, min_cluster=5, max_cluster=20000, kind="ground_boxes", L=4.5),
    "C3L10": dict(n=1_000_000, leaf=0.02, tol=0.1, min_cluster=5, max_cluster=20000, kind="ground_boxes", L=10.0),
    "C5": dict(n=5_000_000, leaf=0.01, tol=0.05, min_cluster=5, max_cluster=20000, kind="dense_volume"),
}


def uniform_field(n, seed):
    rng = np.random.default_rng(seed)
    pts = np.empty((n, 3), dtype=np.float32)
    for a, (lo, hi) in enumerate(FIELD_BOX):
        pts[:, a] = rng.uniform(lo, hi, n).astype(np.float32)
    return pts


def _box_surface(rng, n, centre, side, z0):
    """n points on the 6 faces of an axis-aligned cube [cx±s/2, cy±s/2, z0..z0+s]."""
    face = rng.integers(0, 6, n)
    u = rng.uniform(-0.5, 0.5, n)
    v = rng.uniform(-0.5, 0.5, n)
    p = np.empty((n, 3), dtype=np.float64)
    half = np.where(face % 2 == 0, -0.5, 0.5)
    axis = face // 2
    for a in range(3):
        o1, o2 = [b for b in range(3) if b != a]
        sel = axis == a
        p[sel, a] = half[sel]
        p[sel, o1] = u[sel]
        p[sel, o2] = v[sel]
    p *= side
    p[:, 0] += centre[0]
    p[:, 1] += centre[1]
    p[:, 2] += z0 + side / 2
    return p


def ground_boxes(n, seed, L=4.5, n_boxes=20, tol=0.1):
    """80 % ground plane (z ~ N(0, 3 mm)) on an LxL square + 20 % on the surfaces of
    20 cubes on a jittered 5x4 lattice, hovering 2*tol above the ground so every
    cube is its own connected component."""
    rng = np.random.default_rng(seed)
    n_ground = int(round(0.8 * n))
    n_box = n - n_ground
    g = np.empty((n_ground, 3), dtype=np.float64)
    g[:, 0] = rng.uniform(0.0, L, n_ground)
    g[:, 1] = rng.uniform(0.0, L, n_ground)
    g[:, 2] = rng.normal(0.0, 0.003, n_ground)
    scale = L / 4.5
    side_lo, side_hi = (0.2, 0.4) if L <= 4.5 else (0.3, 0.6)
    nx, ny = 5, 4
    cellx, celly = L / nx, L / ny
    per = np.full(n_boxes, n_box // n_boxes)
    per[: n_box - per.sum()] += 1
    parts = [g]
    b = 0
    for iy in range(ny):
        for ix in range(nx):
            side = rng.uniform(side_lo, side_hi)
            jit = rng.uniform(-0.1, 0.1, 2) * min(scale, 1.5)
            centre = ((ix + 0.5) * cellx + jit[0], (iy + 0.5) * celly + jit[1])
            parts.append(_box_surface(rng, int(per[b]), centre, side, 2.0 * tol))
            b += 1
    pts = np.concatenate(parts, axis=0).astype(np.float32)
    rng.shuffle(pts, axis=0)
    return pts


def dense_volume(n, seed, L=10.0, H=2.0):
    """Dense stress cloud: 60 % uniform in an LxLxH volume + 40 % ground/boxes at scale L.
    Extent/leaf(0.01) = 1000x1000x200 = 2e8 < INT32_MAX, so the overflow rule stays quiet."""
    rng = np.random.default_rng(seed)
    n_vol = int(round(0.6 * n))
    v = np.empty((n_vol, 3), dtype=np.float32)
    v[:, 0] = rng.uniform(0.0, L, n_vol).astype(np.float32)
    v[:, 1] = rng.uniform(0.0, L, n_vol).astype(np.float32)
    v[:, 2] = rng.uniform(0.0, H * 0.95, n_vol).astype(np.float32)
    gb = ground_boxes(n - n_vol, seed + 7919, L=L, tol=0.05)
    pts = np.concatenate([v, gb], axis=0)
    rng.shuffle(pts, axis=0)
    return pts


def make_cloud(config, seed=None, n=None):
    """Cloud of a named config; `n` overrides the size (scaled-down parity cases)."""
    c = CONFIGS[config]
    n = c["n"] if n is None else n
    if seed is None:
        seed = {"C1": 1, "C2": 2, "C3": 3, "C3L10": 3, "C5": 5}[config]
    if c["kind"] == "uniform_field":
        return uniform_field(n, seed)
    if c["kind"] == "ground_boxes":
        return ground_boxes(n, seed, L=c["L"], tol=c["tol"])
    return dense_volume(n, seed)


def params_of(config):
    c = CONFIGS[config]
    return dict(leaf=c["leaf"], tol=c["tol"], min_cluster=c["min_cluster"], max_cluster=c["max_cluster"])
