"""The ROS-free node harness (host/pcop_node, C++ on top of the C ABI) against the oracle: BASELINE config 0
plumbing (params.yaml -> PointCloud2 -> voxel_grid / centroids / euc_clusters / cluster indices)."""
import os
import subprocess

import numpy as np
import pytest

from pointcloud_obstacle_processing_amd import pc2io, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "pointcloud_obstacle_processing_amd", "host")


def read_indices(path):
    lines = open(path).read().split("\n")
    k = int(lines[0])
    return [np.array(lines[1 + c].split()[1:], dtype=np.int32) for c in range(k)]


@pytest.mark.parametrize("mode", ["frame", "stage"])
def test_node_harness_c1(oracle, tmp_path, mode):
    exe = os.path.join(HOST, "pcop_node")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST])
    pts = synth.make_cloud("C1")
    cloud = str(tmp_path / "c1.pc2")
    pc2io.write_pointcloud2(cloud, pts, point_step=32, offsets=(0, 4, 8), width=100, height=100)
    args = [exe, "--params", os.path.join(ROOT, "tests", "golden", "params_c1.yaml"), "--cloud", cloud, "--out", str(tmp_path)]
    if mode == "stage":
        args.append("--stage-api")
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    p = synth.params_of("C1")
    ref = oracle.process_xyz(pts, **p)
    vox = pc2io.read_pointcloud2(str(tmp_path / "voxel_grid.pc2"))
    assert vox["frame_id"] == "world" and vox["point_step"] == 16 and vox["height"] == 1       # od.cpp:290-293
    assert vox["xyz"].tobytes() == ref["voxel_xyz"].tobytes()
    clusters = read_indices(str(tmp_path / "cluster_indices.txt"))
    assert len(clusters) == len(ref["offsets"]) - 1
    for c, idx in enumerate(clusters):
        assert np.array_equal(idx, ref["indices"][ref["offsets"][c]:ref["offsets"][c + 1]])
    if mode == "frame":
        cen = np.loadtxt(str(tmp_path / "centroids.txt"), dtype=np.float32).reshape(-1, 4)
        want = oracle.cluster_summaries(ref["voxel_xyz"], ref["offsets"], ref["indices"])
        assert cen.shape == want.shape and np.max(np.abs(cen - want)) <= 1e-5
        ec = pc2io.read_pointcloud2(str(tmp_path / "euc_clusters.pc2"))
        assert ec["frame_id"] == "/world"                                                      # od.cpp:865
        assert ec["xyz"].tobytes() == ref["voxel_xyz"][ref["indices"]].tobytes()
