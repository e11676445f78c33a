"""Independent NumPy formulation of the RANSAC plane removal, used to cross-check the C++ oracle (oracle/plane_restatement.cpp)
and, through it, the product's host half (csrc/pcop_capi.hip), which is a textual twin of that oracle.  It shares no code with
either: the generator is numpy's own MT19937 (legacy seeding = boost::mt19937's), every float step is a NumPy float32 array
operation with the order written out, the sums are ufunc.accumulate (sequential by definition), the inlier tests are
vectorised.  What it follows is the same reading of PCL - it catches slips of transcription, not a
misreading of PCL; the known-answer tests in tests/test_oracle_planes_crosscheck.py pin the reading where a hand
calculation or numpy.linalg can.
"""
import numpy as np

f32 = np.float32
f64 = np.float64


class BoostMt19937:
    """boost::mt19937(seed) through numpy.random.MT19937 with the classic init_genrand seeding; rnd() = word >> 1
    (boost::uniform_int<>(0, INT_MAX) on a 32-bit engine)."""

    def __init__(self, seed=12345):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(seed)
        self.buf = np.zeros(0, dtype=np.uint64)
        self.pos = 0

    def rnd(self):
        if self.pos >= len(self.buf):
            self.buf = self.bg.random_raw(4096)
            self.pos = 0
        v = int(self.buf[self.pos]) >> 1
        self.pos += 1
        return v


def dot4(c, x, y, z):
    """((a x + b y) + c z) + d in float32, left to right."""
    t = (c[0] * x).astype(f32) + (c[1] * y).astype(f32)
    t = t.astype(f32) + (c[2] * z).astype(f32)
    return (t.astype(f32) + c[3]).astype(f32)


def within(c, pts, thr):
    return np.abs(dot4(c, pts[:, 0], pts[:, 1], pts[:, 2]).astype(f64)) < thr


def valid(c, axis, eps):
    if not eps > 0.0:
        return True
    c = np.asarray(c, dtype=f32)
    d = f64(f32(f32(f32(c[0] * axis[0]) + f32(c[1] * axis[1])) + f32(c[2] * axis[2])))
    n1 = np.sqrt(f64(f32(f32(f32(axis[0] * axis[0]) + f32(axis[1] * axis[1])) + f32(axis[2] * axis[2]))))
    n2 = np.sqrt(f64(f32(f32(f32(c[0] * c[0]) + f32(c[1] * c[1])) + f32(c[2] * c[2]))))
    with np.errstate(all="ignore"):
        rad = d / (n1 * n2)
    rad = min(max(rad, -1.0), 1.0) if rad == rad else rad
    ang = abs(np.arccos(rad))
    ang = min(ang, np.pi - ang)
    return not ang > eps


def model_of(p0, p1, p2):
    with np.errstate(all="ignore"):
        r = (p1 - p0).astype(f32) / (p2 - p0).astype(f32)
        if r[0] == r[1] and r[2] == r[1]:
            return None
        a, b = (p1 - p0).astype(f32), (p2 - p0).astype(f32)
        n = np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]), 0.0], dtype=f32)
        nrm = np.sqrt(f32(f32(f32(f32(n[0] * n[0]) + f32(n[1] * n[1])) + f32(n[2] * n[2])) + f32(n[3] * n[3])))
        n = (n / nrm).astype(f32)
        n[3] = f32(-1.0) * f32(f32(f32(f32(n[0] * p0[0]) + f32(n[1] * p0[1])) + f32(n[2] * p0[2])) + f32(n[3] * f32(1.0)))
    return n


def seq_sum(v):
    """Sequential float32 sum from 0 in array order."""
    v = np.asarray(v, dtype=f32)
    return f32(0.0) if len(v) == 0 else np.add.accumulate(v, dtype=f32)[-1]


def roots2(b, c):
    d = f32(f32(b * b) - f32(f32(4.0) * c))
    d = f32(0.0) if d < 0 else d
    sd = np.sqrt(d)
    return [f32(0.0), f32(f32(0.5) * f32(b - sd)), f32(f32(0.5) * f32(b + sd))]


def roots3(m):
    m = m.astype(f32)
    c0 = f32(f32(f32(f32(f32(m[0, 0] * m[1, 1]) * m[2, 2]) + f32(f32(f32(f32(2.0) * m[0, 1]) * m[0, 2]) * m[1, 2])) - f32(f32(m[0, 0] * m[1, 2]) * m[1, 2]))
             - f32(f32(m[1, 1] * m[0, 2]) * m[0, 2])) - f32(f32(m[2, 2] * m[0, 1]) * m[0, 1])
    c0 = f32(c0)
    c1 = f32(f32(f32(f32(f32(f32(m[0, 0] * m[1, 1]) - f32(m[0, 1] * m[0, 1])) + f32(m[0, 0] * m[2, 2])) - f32(m[0, 2] * m[0, 2])) + f32(m[1, 1] * m[2, 2])) - f32(m[1, 2] * m[1, 2]))
    c2 = f32(f32(m[0, 0] + m[1, 1]) + m[2, 2])
    if abs(c0) < np.finfo(f32).eps:
        return roots2(c2, c1)
    inv3, sqrt3 = f32(f32(1.0) / f32(3.0)), np.sqrt(f32(3.0))
    c2_3 = f32(c2 * inv3)
    a_3 = f32(f32(c1 - f32(c2 * c2_3)) * inv3)
    a_3 = f32(0.0) if a_3 > 0 else a_3
    half_b = f32(f32(0.5) * f32(c0 + f32(c2_3 * f32(f32(f32(f32(2.0) * c2_3) * c2_3) - c1))))
    q = f32(f32(half_b * half_b) + f32(f32(a_3 * a_3) * a_3))
    q = f32(0.0) if q > 0 else q
    rho = np.sqrt(f32(-a_3))
    # (atan2f / cosf / sinf of the C library are correctly rounded for these arguments; NumPy's float32 loops are vectorised
    #  approximations a unit in the last place off now and then: evaluate in double and round once)
    theta = f32(f32(np.arctan2(f64(np.sqrt(f32(-q))), f64(half_b))) * inv3)
    ct, st = f32(np.cos(f64(theta))), f32(np.sin(f64(theta)))
    r = [f32(c2_3 + f32(f32(f32(2.0) * rho) * ct)), f32(c2_3 - f32(rho * f32(ct + f32(sqrt3 * st)))), f32(c2_3 - f32(rho * f32(ct - f32(sqrt3 * st))))]
    if r[0] >= r[1]:
        r[0], r[1] = r[1], r[0]
    if r[1] >= r[2]:
        r[1], r[2] = r[2], r[1]
        if r[0] >= r[1]:
            r[0], r[1] = r[1], r[0]
    if r[0] <= 0:
        return roots2(c2, c1)
    return r


def refit(pts, inl, c_in, axis, eps):
    c_in = np.asarray(c_in, dtype=f32)
    if not valid(c_in, axis, eps) or len(inl) <= 3:
        return c_in.copy()
    q = pts[inl]
    x, y, z = q[:, 0], q[:, 1], q[:, 2]
    cnt = f32(len(inl))
    acc = [f32(seq_sum(v) / cnt) for v in ((x * x).astype(f32), (x * y).astype(f32), (x * z).astype(f32), (y * y).astype(f32), (y * z).astype(f32), (z * z).astype(f32), x, y, z)]
    m = np.zeros((3, 3), dtype=f32)
    m[0, 0] = acc[0] - f32(acc[6] * acc[6]); m[0, 1] = acc[1] - f32(acc[6] * acc[7]); m[0, 2] = acc[2] - f32(acc[6] * acc[8])
    m[1, 1] = acc[3] - f32(acc[7] * acc[7]); m[1, 2] = acc[4] - f32(acc[7] * acc[8]); m[2, 2] = acc[5] - f32(acc[8] * acc[8])
    m[1, 0], m[2, 0], m[2, 1] = m[0, 1], m[0, 2], m[1, 2]
    scale = np.max(np.abs(m))
    if scale <= np.finfo(f32).tiny:
        scale = f32(1.0)
    s = (m / scale).astype(f32)
    ev = roots3(s)
    s = s.copy()
    for a in range(3):
        s[a, a] = f32(s[a, a] - ev[0])

    def cross(u, v):
        return np.array([f32(u[1] * v[2]) - f32(u[2] * v[1]), f32(u[2] * v[0]) - f32(u[0] * v[2]), f32(u[0] * v[1]) - f32(u[1] * v[0])], dtype=f32)

    def sq(v):
        return f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))

    v1, v2, v3 = cross(s[0], s[1]), cross(s[0], s[2]), cross(s[1], s[2])
    l1, l2, l3 = sq(v1), sq(v2), sq(v3)
    if l1 >= l2 and l1 >= l3:
        best, ln = v1, l1
    elif l2 >= l1 and l2 >= l3:
        best, ln = v2, l2
    else:
        best, ln = v3, l3
    with np.errstate(all="ignore"):
        nrm = (best / np.sqrt(ln)).astype(f32)
    c = np.array([nrm[0], nrm[1], nrm[2], 0.0], dtype=f32)
    c[3] = f32(-1.0) * f32(f32(f32(f32(c[0] * acc[6]) + f32(c[1] * acc[7])) + f32(c[2] * acc[8])) + f32(f32(0.0) * f32(1.0)))
    return c if valid(c, axis, eps) else c_in.copy()


def ransac_k(count, n, probability=0.99):
    """RandomSampleConsensus's iteration budget after a model with `count` inliers of n."""
    w = f64(count) * (1.0 / f64(n))
    p_no = 1.0 - w ** 3.0
    p_no = min(max(np.finfo(f64).eps, p_no), 1.0 - np.finfo(f64).eps)
    return np.log(1.0 - probability) / np.log(p_no)


def segment(pts, thr, axis, eps, max_iterations=50, probability=0.99):
    """One seg.segment(): (inlier indices or None, coefficients, iterations)."""
    n = len(pts)
    if n < 3:
        return None, None, 0
    gen = BoostMt19937(12345)
    shuffled = list(range(n))
    iterations, best_count, k, skipped = 0, -2 ** 31 + 1, 1.0, 0
    best = None
    while iterations < k and skipped < 10 * max_iterations:
        good = False
        for _ in range(1000):
            for i in range(3):
                j = i + gen.rnd() % (n - i)
                shuffled[i], shuffled[j] = shuffled[j], shuffled[i]
            p0, p1, p2 = pts[shuffled[0]], pts[shuffled[1]], pts[shuffled[2]]
            with np.errstate(all="ignore"):
                r = (p1 - p0).astype(f32) / (p2 - p0).astype(f32)
            if (r[0] != r[1]) or (r[2] != r[1]):
                good = True
                break
        if not good:
            break
        c = model_of(p0, p1, p2)
        if c is None:
            skipped += 1
            continue
        cnt = int(within(c, pts, thr).sum()) if valid(c, axis, eps) else 0
        if cnt > best_count:
            best_count, best = cnt, c
            k = ransac_k(cnt, n, probability)
        iterations += 1
        if iterations > max_iterations:
            break
    if best is None:
        return None, None, iterations
    inl = np.flatnonzero(within(best, pts, thr)) if valid(best, axis, eps) else np.zeros(0, dtype=np.int64)
    refined = refit(pts, inl, best, axis, eps)
    fin = np.flatnonzero(within(refined, pts, thr)) if valid(refined, axis, eps) else np.zeros(0, dtype=np.int64)
    return fin, refined, iterations


def segment_planes(xyz, dist_thresh, axis=(0.0, 0.0, 1.0), eps_angle=20.0):
    """segment_plane_and_extract_indices (od.cpp:342-428): dict(xyz, last_plane, cloud_f, coeffs, iterations, n_planes, found_all)."""
    cloud = np.ascontiguousarray(xyz, dtype=f32).reshape(-1, 3)
    axis = np.asarray(axis, dtype=f32)
    thr = f64(f32(dist_thresh))
    nr = len(cloud)
    last = np.zeros((0, 3), dtype=f32)
    before = None
    coeffs, its = [], []
    found_all = True
    while len(cloud) > 0.3 * nr:
        inl, c, it = segment(cloud, thr, axis, float(eps_angle))
        its.append(it)
        if inl is None or len(inl) == 0:
            found_all = False
            its.pop() if inl is None and False else None
            break
        mask = np.zeros(len(cloud), dtype=bool)
        mask[inl] = True
        last, before = cloud[mask], cloud
        cloud = cloud[~mask]
        coeffs.append(c)
    return dict(xyz=cloud, last_plane=last, cloud_f=before, coeffs=np.array(coeffs, dtype=f32).reshape(-1, 4), iterations=its[:len(coeffs)],
                n_planes=len(coeffs), found_all=found_all)
