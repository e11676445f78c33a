"""CPU tests of the statistical-outlier-removal oracle (oracle/sor_restatement.cpp, restating
pcl::StatisticalOutlierRemoval as remove_statistical_outliers uses it, od.cpp:316-340): brute force, a case by hand."""
import numpy as np
import pytest

f32 = np.float32


def brute(pts, mean_k, mul):
    """Independent formulation: all pairs in float32 ((dx^2 + dy^2) + dz^2), sort, sqrt in double, sequential sums."""
    pts = np.asarray(pts, dtype=f32)
    fin = np.isfinite(pts).all(axis=1)
    q = pts[fin]
    d = ((q[:, None, 0] - q[None, :, 0]) ** 2 + (q[:, None, 1] - q[None, :, 1]) ** 2) + (q[:, None, 2] - q[None, :, 2]) ** 2
    s = np.sort(d.astype(f32), axis=1)[:, 1:mean_k + 1].astype(np.float64)
    acc = np.zeros(len(q))
    for t in range(mean_k):            # ascending, one add at a time
        acc = acc + np.sqrt(s[:, t])
    dist = np.zeros(len(pts), dtype=f32)
    dist[fin] = (acc / mean_k).astype(f32)
    total, sq = 0.0, 0.0
    for v in dist:                      # sequential, in index order; the square is a float product
        total += float(v)
        sq += float(f32(v) * f32(v))
    nv = int(fin.sum())
    mean = total / nv
    var = (sq - total * total / nv) / (nv - 1.0)
    thr = mean + float(f32(mul)) * np.sqrt(var)
    return dist, thr, ~(dist.astype(np.float64) > thr)


@pytest.mark.parametrize("seed,n,k,mul", [(1, 600, 15, 1.0), (2, 900, 4, 0.5), (3, 700, 31, 4.0)])
def test_matches_brute_force(oracle, seed, n, k, mul):
    rng = np.random.default_rng(seed)
    pts = np.concatenate([rng.normal(0, 0.3, (n, 3)), rng.uniform(-3, 3, (n // 10, 3))]).astype(f32)   # a blob + stragglers
    pts[::97, 2] = np.nan
    r = oracle.sor(pts, k, mul)
    dist, thr, keep = brute(pts, k, mul)
    assert r["distances"].tobytes() == dist.tobytes()
    assert r["threshold"] == thr
    assert np.array_equal(r["keep"], keep)
    assert 0 < keep.sum() < len(pts)
    assert keep[::97].all()            # non-finite points: distance 0, never above the threshold, stay in (PCL semantics)


def test_by_hand(oracle):
    # five points on a line at 0,1,2,3 and 10; mean_k = 2: mean distances 1.5, 1, 1, 1.5, 7.5
    pts = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [10, 0, 0]], dtype=f32)
    r = oracle.sor(pts, 2, 1.0)
    assert list(r["distances"]) == [1.5, 1.0, 1.0, 1.5, 7.5]
    # mean 2.5, variance (3.0625*2... ) -> sample sigma = sqrt(((1)^2+(1.5)^2+(1.5)^2+(1)^2+(5)^2)/4) = sqrt(7.875)
    assert abs(r["threshold"] - (2.5 + np.sqrt(7.875))) < 1e-12
    assert list(r["keep"]) == [True, True, True, True, False]


def test_too_few_points_is_refused(oracle):
    with pytest.raises(ValueError):
        oracle.sor(np.zeros((10, 3), dtype=f32), 15, 1.0)
