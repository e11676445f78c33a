"""CPU restatement of the RANSAC plane removal (oracle/plane_restatement.cpp; od.cpp:342-428): known answers.
PARITY UNPINNED like the rest of the PCL path: PCL is not installed and the reference holds no fixtures; these tests pin
the pieces that have an independent definition (the Mersenne twister, the plane through three points, a least-squares
plane) and the loop's behaviour on hand-made clouds."""
import numpy as np

f32 = np.float32


def test_generator_is_the_mersenne_twister_seeded_12345(oracle):
    """boost::mt19937 seeded with 12345u, uniform_int<>(0, INT_MAX) = the 32-bit word / 2: checked against NumPy's MT19937
    with the same legacy seeding."""
    bg = np.random.MT19937()
    bg._legacy_seeding(12345)
    assert np.array_equal(oracle.plane_rnd(2000).astype(np.int64), (bg.random_raw(2000) >> 1).astype(np.int64))


def plane_cloud(n, normal, d, noise, seed, lo=-2.0, hi=2.0):
    rng = np.random.default_rng(seed)
    nrm = np.asarray(normal, dtype=np.float64) / np.linalg.norm(normal)
    a = np.cross(nrm, [1.0, 0.3, 0.2]); a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    uv = rng.uniform(lo, hi, (n, 2))
    pts = uv[:, :1] * a + uv[:, 1:] * b - d * nrm + rng.normal(0, noise, (n, 1)) * nrm
    return pts.astype(f32)


def test_single_plane_is_found_and_refitted(oracle):
    pts = plane_cloud(5000, (0.1, -0.2, 1.0), 0.35, 0.002, 1)
    r = oracle.segment_planes(pts, 0.02)
    assert r["n_planes"] == 1 and r["found_all"] and len(r["xyz"]) == 0 and len(r["last_plane"]) == 5000
    c = r["coeffs"][0].astype(np.float64)
    nrm = np.array([0.1, -0.2, 1.0]) / np.linalg.norm([0.1, -0.2, 1.0])
    s = np.sign(c[:3] @ nrm)
    assert np.allclose(s * c[:3], nrm, atol=2e-3) and abs(s * c[3] - 0.35) < 2e-3 and abs(np.linalg.norm(c[:3]) - 1) < 1e-5
    # the refit is the least-squares plane of the inliers: compare with an SVD in double
    ctr = pts.astype(np.float64).mean(axis=0)
    _, _, vt = np.linalg.svd(pts.astype(np.float64) - ctr)
    assert abs(abs(vt[2] @ c[:3]) - 1.0) < 1e-4


def test_loop_removes_planes_until_thirty_percent_is_left(oracle):
    """60 % ground, 25 % a tilted wall, 15 % scattered: two planes go, the scatter (15 % < 30 %) stays."""
    ground = plane_cloud(6000, (0, 0, 1), 0.0, 0.003, 2)
    wall = plane_cloud(2500, (1, 0.2, 0.1), 1.0, 0.003, 3)
    rng = np.random.default_rng(4)
    blob = rng.uniform(-2, 2, (1500, 3)).astype(f32) + f32([0, 0, 3])
    pts = np.concatenate([ground, wall, blob])[rng.permutation(10000)]
    r = oracle.segment_planes(pts, 0.02)
    assert r["n_planes"] == 2 and r["found_all"]
    assert abs(abs(r["coeffs"][0][2]) - 1.0) < 1e-3                      # the larger plane first: the ground
    assert 1400 <= len(r["xyz"]) <= 1600 and len(r["xyz"]) <= 0.3 * len(pts)
    # what is left keeps the input order
    left = r["xyz"]
    idx = [np.flatnonzero((pts == p).all(axis=1))[0] for p in left[:50]]
    assert idx == sorted(idx)


def test_no_plane_in_too_few_points(oracle):
    r = oracle.segment_planes(np.array([[0, 0, 0], [1, 0, 0]], dtype=f32), 0.02)
    assert r["n_planes"] == 0 and not r["found_all"] and len(r["xyz"]) == 2
    r = oracle.segment_planes(np.zeros((0, 3), dtype=f32), 0.02)
    assert r["n_planes"] == 0 and r["found_all"] and len(r["xyz"]) == 0     # 0 > 0.3 * 0 is false: the loop never runs


def test_perpendicular_constraint_in_radians(oracle):
    """eps_angle > 0 rejects planes whose normal is further than eps_angle from the axis; the node's 20 (an int read as
    radians, od.cpp:111, 371) accepts everything, a real 0.1 rad rejects a wall."""
    wall = plane_cloud(4000, (1, 0, 0), 0.5, 0.002, 5)
    assert oracle.segment_planes(wall, 0.02, eps_angle=20.0)["n_planes"] == 1
    r = oracle.segment_planes(wall, 0.02, eps_angle=0.1)
    assert r["n_planes"] == 0 and not r["found_all"] and len(r["xyz"]) == 4000
    assert oracle.segment_planes(wall, 0.02, axis=(1, 0, 0), eps_angle=0.1)["n_planes"] == 1
