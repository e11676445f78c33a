"""What PCL leaves open, and what it cannot change (SURVEY.md 8a-1 steps 6 and 8, 8c "margin diagnostic").

The oracle fixes one legal instance of pcl::VoxelGrid's implementation-defined parts: equal voxel keys keep ascending
point index (PCL's std::sort is unstable) and the centroid is sum / float(n) (PCL >= 1.8).  The node's era is PCL 1.7.2,
whose Eigen 3.2 accumulator multiplies by the reciprocal instead, and any intra-voxel order is legal.  These tests show,
on the committed configurations, that every such variant gives the same voxels in the same order, centroids within the
north star's 1e-4, and - the part that matters downstream - identical cluster offsets and indices; and that no pair of
centroids sits within 4 ulp of the squared radius in a position where it would decide a cluster border.
PARITY UNPINNED all the same: none of this is PCL itself (it is not installed and the reference holds no fixtures)."""
import glob
import os

import numpy as np
import pytest

from pointcloud_obstacle_processing_amd import synth

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c*_seed*.npz")))
CASES = [("C1", None, 1), ("C2", 20000, 2), ("C3", 60000, 3), ("C3", 400_000, 4)]


@pytest.mark.parametrize("cfg,n,seed", CASES)
def test_cluster_indices_do_not_depend_on_pcl_version_or_sort_order(oracle, cfg, n, seed):
    p = synth.params_of(cfg)
    pts = synth.make_cloud(cfg, seed=seed, n=n)
    base, pt = oracle.voxel_downsample(pts, p["leaf"])
    off0, idx0 = oracle.euclidean_clusters(base, p["tol"], p["min_cluster"], p["max_cluster"])
    for variant in (oracle.VARIANT_RECIPROCAL, oracle.VARIANT_DESCENDING, oracle.VARIANT_RECIPROCAL | oracle.VARIANT_DESCENDING,
                    oracle.VARIANT_STDSORT, oracle.VARIANT_STDSORT | oracle.VARIANT_RECIPROCAL):
        vox, pt_v = oracle.voxel_downsample(pts, p["leaf"], variant=variant)
        assert pt_v == pt and vox.shape == base.shape                       # same voxels, same order
        assert np.max(np.abs(vox - base), initial=0.0) < 1e-4               # the north star's centroid bound
        off, idx = oracle.euclidean_clusters(vox, p["tol"], p["min_cluster"], p["max_cluster"])
        assert np.array_equal(off, off0) and np.array_equal(idx, idx0), "variant %d moves a cluster border" % variant
    # the reciprocal really is a different rounding somewhere (otherwise this test shows nothing)
    vox1, _ = oracle.voxel_downsample(pts, p["leaf"], variant=oracle.VARIANT_RECIPROCAL)
    assert vox1.tobytes() != base.tobytes() or cfg == "C1"


def test_the_order_std_sort_itself_leaves_is_neither_ascending_nor_harmful(oracle):
    """PCL's own call - std::sort comparing the voxel key alone - with this image's libstdc++: the order of a voxel's points is then
    what introsort leaves, neither the canonical ascending one (which the GPU path reproduces) nor its reverse.  The sums differ in
    their last bits from the canonical ones in some voxels (so the variant is a real one), never by the north star's 1e-4, and the
    voxels, their order and every cluster index are the same."""
    p = synth.params_of("C3")
    pts = synth.make_cloud("C3", seed=11, n=300_000)
    base, _ = oracle.voxel_downsample(pts, p["leaf"])
    lit, _ = oracle.voxel_downsample(pts, p["leaf"], variant=oracle.VARIANT_STDSORT)
    desc, _ = oracle.voxel_downsample(pts, p["leaf"], variant=oracle.VARIANT_DESCENDING)
    assert lit.shape == base.shape
    differs = np.any(lit != base, axis=1)
    assert 0 < int(differs.sum()) < len(base)                 # some voxels' sums round differently, not all (single-point voxels cannot)
    assert lit.tobytes() != desc.tobytes()
    assert float(np.max(np.abs(lit - base))) < 1e-4
    # a last-bit difference at most: the same additions in another order (a voxel of C3 holds a handful of points)
    ulp = np.spacing(np.maximum(np.abs(base), np.float32(1e-3)).astype(np.float32))
    assert float(np.max(np.abs(lit - base) / ulp)) <= 4.0
    off0, idx0 = oracle.euclidean_clusters(base, p["tol"], p["min_cluster"], p["max_cluster"])
    off1, idx1 = oracle.euclidean_clusters(lit, p["tol"], p["min_cluster"], p["max_cluster"])
    assert np.array_equal(off0, off1) and np.array_equal(idx0, idx1)


def test_knife_edge_report_equals_brute_force(oracle):
    rng = np.random.default_rng(17)
    pts = rng.uniform(0, 1, (1500, 3)).astype(np.float32)
    pts[100] = pts[7] + np.array([0.125, 0, 0], dtype=np.float32)          # d2 = r2 exactly for tol 0.125: a knife-edge pair
    for tol in (0.125, 0.1, 0.05):
        knife, critical = oracle.knife_edge_report(pts, tol, 4)
        assert knife == oracle.knife_edge_pairs(pts, tol, 4)
        assert critical <= knife
    assert oracle.knife_edge_report(pts, 0.125, 4)[0] >= 1


def test_knife_edge_pair_that_decides_a_border_is_reported(oracle):
    """Two far-apart pairs of points; one pair at exactly the radius (strictly-less fails: two singletons), so a last-bit
    change would merge them: critical."""
    pts = np.array([[0, 0, 0], [0.5, 0, 0], [10, 0, 0], [10.2, 0, 0]], dtype=np.float32)
    knife, critical = oracle.knife_edge_report(pts, 0.5, 4)
    assert (knife, critical) == (1, 1)
    assert oracle.knife_edge_report(pts, 0.6, 4) == (0, 0)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_committed_fixtures_have_no_deciding_knife_edge(oracle, path):
    g = np.load(path, allow_pickle=False)
    knife, critical = oracle.knife_edge_report(g["voxel_xyz"], float(g["tol"]), 4)
    assert (knife, critical) == (int(g["knife_edge_pairs"]), int(g["knife_edge_critical"]))
    assert critical == 0
