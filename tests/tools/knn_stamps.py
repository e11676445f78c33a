"""Outlier removal on the voxel cloud of a field frame with strays: stage times of the call, the points the first k-NN pass
(the 27 cells around a point) leaves to the second (one wave per point)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth
from test_gpu_node_chain import field_cloud
pts = field_cloud(1_000_000, 77)
cloud = O.occupancy(pts, O.occ_params(0.0, 4.5, 0.0, 3.78, -0.5, 0.6, 0.15, 0.5))["xyz"]
for name, vox in (("field+strays", O.voxel_downsample(cloud, 0.02)[0]), ("C3", O.voxel_downsample(synth.make_cloud("C3", seed=3000), 0.02)[0])):
    c = capi.Context(len(vox), profiling=True)
    for _ in range(3):
        got = c.statistical_outlier_removal(vox, 15, 1.0)
    left = int(c.debug_buffer("bin_total", np.uint32, 1)[0])
    print(name, "points", len(vox), "left to k_knn_far", left, {nm: round(1e3 * ms, 1) for nm, ms in c.stage_times()})
    ref = O.sor(vox, 15, 1.0)
    print("   bit-exact:", got["distances"].tobytes() == ref["distances"].tobytes() and got["threshold"] == ref["threshold"])
    c.close()
