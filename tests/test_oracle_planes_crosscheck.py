"""The RANSAC plane removal: the C++ oracle (oracle/plane_restatement.cpp) against an independent NumPy formulation
(tests/np_planes.py), and known answers that do not go through either's text - numpy.linalg for the refit, hand
arithmetic for the iteration budget, stated outcomes for the degenerate clouds.  (The product's host half is a textual
twin of the oracle: the GPU tests compare those two, this file is what stands behind the oracle.  Parity with PCL itself
stays unpinned: SURVEY.md 8c.)"""
import numpy as np
import pytest

from tests import np_planes as NP

f32 = np.float32


def tilted_plane(rng, n, normal, d, noise, extent=2.0):
    """n points near the plane normal . p + d = 0."""
    normal = np.asarray(normal, dtype=np.float64)
    normal = normal / np.linalg.norm(normal)
    a = np.cross(normal, [1.0, 0.3, 0.2]); a /= np.linalg.norm(a)
    b = np.cross(normal, a)
    uv = rng.uniform(-extent, extent, (n, 2))
    p = -d * normal + uv[:, :1] * a + uv[:, 1:] * b + rng.normal(0, noise, (n, 1)) * normal
    return p.astype(f32)


def clouds():
    rng = np.random.default_rng(77)
    yield "ground+clutter", np.concatenate([tilted_plane(rng, 1500, (0, 0, 1), -0.1, 0.004), rng.uniform(-2, 2, (500, 3)).astype(f32)]), 0.02
    yield "two planes", np.concatenate([tilted_plane(rng, 1200, (0.05, -0.02, 1), 0.3, 0.003), tilted_plane(rng, 700, (0.9, 0.1, 0.2), -1.0, 0.003),
                                        rng.uniform(-2, 2, (150, 3)).astype(f32)]), 0.015
    yield "no plane worth the name", rng.uniform(-1, 1, (400, 3)).astype(f32), 0.001
    yield "a line", np.stack([np.linspace(0, 1, 50), np.zeros(50), np.zeros(50)], axis=1).astype(f32), 0.01
    q = np.concatenate([tilted_plane(rng, 900, (0, 0.1, 1), 0.0, 0.002), rng.uniform(-1, 1, (200, 3)).astype(f32)])
    q[::41] = np.nan
    yield "NaN rows", q, 0.01
    yield "two points", np.array([[0, 0, 0], [1, 1, 1]], dtype=f32), 0.01


@pytest.mark.parametrize("name,pts,thr", list(clouds()), ids=[c[0] for c in clouds()])
def test_oracle_equals_numpy_formulation(oracle, name, pts, thr):
    a = oracle.segment_planes(pts, thr)
    b = NP.segment_planes(pts, thr)
    assert a["n_planes"] == b["n_planes"] and a["found_all"] == b["found_all"]
    assert a["xyz"].tobytes() == b["xyz"].tobytes()
    assert a["last_plane"].tobytes() == b["last_plane"].tobytes()
    assert a["coeffs"].tobytes() == b["coeffs"].tobytes()
    assert list(a["iterations"]) == list(b["iterations"])


def test_generator_is_boost_mt19937(oracle):
    """The first values of rnd() after seed 12345: the oracle's own twister against numpy's (legacy seeding), and the
    published first output of mt19937 seeded with 5489 as the anchor outside both."""
    g = NP.BoostMt19937(12345)
    assert list(oracle.plane_rnd(64)) == [g.rnd() for _ in range(64)]
    ref = np.random.MT19937()
    ref._legacy_seeding(5489)
    assert int(ref.random_raw(1)[0]) == 3499211612          # the reference value of MT19937's first word (Matsumoto & Nishimura)


def test_refit_agrees_with_numpy_linalg(oracle):
    """The least-squares refit (computeMeanAndCovarianceMatrix + eigen33's closed form) against numpy.linalg.eigh of the
    float64 covariance: the normal within 1e-4, d within 1e-4 (float32 sums of a thousand products)."""
    rng = np.random.default_rng(5)
    for normal, d in (((0, 0, 1), -0.25), ((0.1, -0.2, 1), 0.4), ((0.02, 0.01, 1), -1.5)):
        pts = tilted_plane(rng, 1000, normal, d, 0.002)
        r = oracle.segment_planes(pts, 0.02)
        assert r["n_planes"] >= 1
        c = r["coeffs"][0].astype(np.float64)
        q = pts.astype(np.float64)
        w, v = np.linalg.eigh(np.cov(q.T, bias=True))
        n = v[:, 0]
        n = n if np.dot(n, c[:3]) > 0 else -n
        assert np.max(np.abs(n - c[:3])) < 1e-4
        assert abs(c[3] + np.dot(n, q.mean(axis=0))) < 1e-4
        assert abs(np.linalg.norm(c[:3]) - 1.0) < 1e-6


def test_exact_plane_by_hand(oracle):
    """Every point on z = 0.5 exactly (0.5 and the lattice coordinates are exact in float32): whichever three points the generator draws,
    the model is (0, 0, +-1, -+0.5) exactly, every point is an inlier, one iteration is enough (w = 1: k = log(0.01) / log(eps) < 1),
    the refit of exact data gives the same plane, the whole cloud goes in one round and nothing is left."""
    g = np.arange(-8, 9, dtype=f32) * f32(0.125)
    x, y = np.meshgrid(g, g)
    pts = np.stack([x.ravel(), y.ravel(), np.full(x.size, 0.5, dtype=f32)], axis=1)
    r = oracle.segment_planes(pts, 0.01)
    assert r["n_planes"] == 1 and len(r["xyz"]) == 0 and len(r["last_plane"]) == len(pts) and r["found_all"]
    assert list(r["iterations"]) == [1]
    c = r["coeffs"][0]
    assert abs(abs(c[2]) - 1.0) < 1e-6 and abs(c[0]) < 1e-6 and abs(c[1]) < 1e-6 and abs(c[3] + 0.5 * np.sign(c[2])) < 1e-6


def test_iteration_budget_formula():
    """k = log(1 - 0.99) / log(1 - w^3), w clamped through DBL_EPSILON: the values a hand calculation gives."""
    assert abs(NP.ransac_k(50, 100) - np.log(0.01) / np.log(1 - 0.125)) < 1e-12      # w = 0.5: 34.49 iterations
    assert 34 < NP.ransac_k(50, 100) < 35
    assert NP.ransac_k(100, 100) < 1.0                                                 # w = 1: p_no = eps, one iteration
    assert NP.ransac_k(0, 100) > 1e15                                                  # w = 0: p_no = 1 - eps, the 50-iteration cap decides
    assert abs(NP.ransac_k(90, 100) - np.log(0.01) / np.log(1 - 0.729)) < 1e-12       # 3.53


def test_degenerate_clouds_by_statement(oracle):
    """A line: isSampleGood fails a thousand times, "No samples could be selected": no model, the cloud stays as it is.
    Fewer than three points: no model either.  The 30 % rule: a cloud whose first plane takes more than 70 % stops after it."""
    line = np.stack([np.linspace(0, 1, 50), np.zeros(50), np.zeros(50)], axis=1).astype(f32)
    r = oracle.segment_planes(line, 0.01)
    assert r["n_planes"] == 0 and not r["found_all"] and r["xyz"].tobytes() == line.tobytes() and len(r["last_plane"]) == 0
    r = oracle.segment_planes(line[:2], 0.01)
    assert r["n_planes"] == 0 and not r["found_all"] and len(r["xyz"]) == 2
    rng = np.random.default_rng(3)
    pts = np.concatenate([tilted_plane(rng, 900, (0, 0, 1), 0.0, 0.001), rng.uniform(-2, 2, (100, 3)).astype(f32) + f32(5.0)])
    r = oracle.segment_planes(pts, 0.01)
    assert r["n_planes"] == 1 and r["found_all"] and len(r["xyz"]) <= 0.3 * len(pts)
