"""Known-answer tests pinning each [EXT] rule the oracle restates.

The reference repository has no tests or fixtures (parity unpinned), so every
expected value here is hand-computable from the published PCL/FLANN semantics.
"""
import numpy as np
import pytest

f32 = np.float32


def test_kat1_strict_radius(oracle):
    # d2 == r2 exactly (0.25) => NOT neighbours (FLANN RadiusResultSet: dist < radius)
    pts = np.array([[0, 0, 0], [0.5, 0, 0]], dtype=f32)
    off, idx = oracle.euclidean_clusters(pts, 0.5, 1, 10)
    assert list(np.diff(off)) == [1, 1]
    tol_up = np.nextafter(f32(0.5), f32(1.0))
    off, idx = oracle.euclidean_clusters(pts, tol_up, 1, 10)
    assert list(np.diff(off)) == [2] and list(idx) == [0, 1]
    # both search back-ends agree
    off2, idx2 = oracle.euclidean_clusters(pts, 0.5, 1, 10, use_tree=False)
    assert list(np.diff(off2)) == [1, 1]


def test_kat2_output_order_x_fastest(oracle):
    # three voxels handed over in descending key order; output ascends by ix + iy*dx + iz*dx*dy
    pts = np.array([[0.01, 0.01, 0.11],   # (0,0,2)
                    [0.01, 0.06, 0.01],   # (0,1,0)
                    [0.06, 0.01, 0.01],   # (1,0,0)
                    [0.01, 0.01, 0.01]],  # (0,0,0)
                   dtype=f32)
    vox, pt = oracle.voxel_downsample(pts, 0.05)
    assert not pt
    assert np.array_equal(vox, pts[[3, 2, 1, 0]])
    keys, grid, _ = oracle.voxel_keys(pts, 0.05)
    assert list(grid[3:6]) == [2, 2, 3]          # div_b
    assert list(grid[6:9]) == [1, 2, 4]          # divb_mul
    assert list(keys) == [8, 2, 1, 0]


def test_kat3_negative_coordinates(oracle):
    pts = np.array([[-0.01, 0, 0], [0.01, 0, 0]], dtype=f32)
    keys, grid, _ = oracle.voxel_keys(pts, 0.05)
    assert grid[0] == -1                         # min_b.x = floor(-0.2) = -1
    assert list(keys) == [0, 1]                  # different voxels
    vox, _ = oracle.voxel_downsample(pts, 0.05)
    assert vox.shape[0] == 2


def test_kat4_centroid_float32_sequential(oracle):
    pts = np.array([[0.01, 0, 0], [0.02, 0, 0], [0.04, 0, 0]], dtype=f32)
    vox, _ = oracle.voxel_downsample(pts, 0.05)
    expect = ((f32(0) + f32(0.01)) + f32(0.02) + f32(0.04)) / f32(3.0)
    assert vox.shape == (1, 3)
    assert vox[0, 0].tobytes() == f32(expect).tobytes()
    # order matters in float32: the canonical order is ascending point index
    pts2 = np.array([[16777216.0, 0, 0], [1.0, 0, 0], [1.0, 0, 0]], dtype=f32)
    vox2, _ = oracle.voxel_downsample(pts2, 1e9)
    assert vox2[0, 0] == f32(16777216.0) / f32(3.0)   # the two +1 are absorbed one at a time


def test_kat5_overflow_rule_passthrough(oracle):
    pts = np.array([[0, 0, 0], [3000, 0, 0], [0, 3000, 0], [0, 0, 3000]], dtype=f32)
    vox, pt = oracle.voxel_downsample(pts, 0.001)
    assert pt and np.array_equal(vox, pts)
    # one axis only: 3e6 cells < INT32_MAX => no passthrough
    pts1 = np.array([[0, 0, 0], [3000, 0, 0]], dtype=f32)
    vox1, pt1 = oracle.voxel_downsample(pts1, 0.001)
    assert not pt1 and vox1.shape[0] == 2


def test_kat6_size_filter_no_bfs_truncation(oracle):
    tol = f32(0.1)
    chain = np.zeros((7, 3), dtype=f32)
    chain[:, 0] = np.arange(7, dtype=f32) * f32(0.09)
    off, idx = oracle.euclidean_clusters(chain, tol, 5, 6)
    assert len(off) - 1 == 0                     # the component of 7 is dropped whole
    off, idx = oracle.euclidean_clusters(chain, tol, 5, 7)
    assert list(np.diff(off)) == [7] and list(idx) == list(range(7))


def test_kat7_cluster_ordering(oracle):
    # discovery sizes [5,9,5,7] -> 9, 7, then the 5s as libstdc++ sort on the reverse range leaves them
    order = oracle.cluster_order([5, 9, 5, 7])
    assert list(order[:2]) == [1, 3]
    assert sorted(order[2:]) == [0, 2]
    # end-to-end: four separated chains of those sizes, discovered in index order
    tol = f32(0.1)
    pts, sizes = [], [5, 9, 5, 7]
    for c, s in enumerate(sizes):
        for j in range(s):
            pts.append([j * 0.05, c * 1.0, 0.0])
    pts = np.array(pts, dtype=f32)
    off, idx = oracle.euclidean_clusters(pts, tol, 5, 100)
    starts = np.cumsum([0] + sizes)
    got = [int(np.searchsorted(starts, idx[off[c]], side="right") - 1) for c in range(4)]
    assert got == list(order)
    for c in range(4):
        seg = idx[off[c]:off[c + 1]]
        assert np.all(np.diff(seg) > 0)          # ascending inside a cluster


def test_kat8_r2_rounding(oracle):
    tol = f32(0.1)
    expect = f32(np.float64(tol) * np.float64(tol))
    assert oracle.r2(tol).tobytes() == expect.tobytes()
    assert oracle.r2(tol).view(np.uint32) == np.uint32(0x3C23D70B)


def test_empty_and_single(oracle):
    vox, pt = oracle.voxel_downsample(np.zeros((0, 3), dtype=f32), 0.05)
    assert vox.shape == (0, 3) and not pt
    off, idx = oracle.euclidean_clusters(np.zeros((0, 3), dtype=f32), 0.1, 1, 10)
    assert list(off) == [0]
    off, idx = oracle.euclidean_clusters(np.array([[1, 2, 3]], dtype=f32), 0.1, 1, 10)
    assert list(off) == [0, 1] and list(idx) == [0]


def test_nan_points_dropped_when_not_dense(oracle):
    pts = np.array([[0.01, 0.01, 0.01], [np.nan, 0, 0], [0.02, 0.01, 0.01], [0, np.inf, 0]], dtype=f32)
    vox, pt = oracle.voxel_downsample(pts, 0.05, is_dense=False)
    assert vox.shape == (1, 3)
    assert vox[0, 0] == (f32(0.01) + f32(0.02)) / f32(2)


def test_layouts_agree(oracle):
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (500, 3)).astype(f32)
    ref, _ = oracle.voxel_downsample(pts, 0.1, layout=oracle.LAYOUT_AOS12)
    for lay in (oracle.LAYOUT_AOS16, oracle.LAYOUT_SOA):
        got, _ = oracle.voxel_downsample(pts, 0.1, layout=lay)
        assert np.array_equal(ref, got)
    o0, i0 = oracle.euclidean_clusters(ref, 0.25, 2, 1000)
    for lay in (oracle.LAYOUT_AOS16, oracle.LAYOUT_SOA):
        o1, i1 = oracle.euclidean_clusters(ref, 0.25, 2, 1000, layout=lay)
        assert np.array_equal(o0, o1) and np.array_equal(i0, i1)


def test_duplicate_points_are_connected(oracle):
    # exact duplicates: FLANN sorts results by (distance, index), so the skipped element 0 is
    # always the smallest-index duplicate and the component is still found whole
    pts = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [5, 5, 5]], dtype=f32)
    off, idx = oracle.euclidean_clusters(pts, 0.1, 1, 10)
    assert list(np.diff(off)) == [3, 1] and list(idx) == [0, 1, 2, 3]
