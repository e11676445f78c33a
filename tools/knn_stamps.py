"""Diagnostic build only (make HIPFLAGS+=-DPCOP_STAMPS): per wave of k_knn_mean, the time of its three parts (the 27 cells,
the shells row by row, the blocks) on the voxel cloud of a field frame with strays."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi
from test_gpu_node_chain import field_cloud
pts = field_cloud(1_000_000, 77)
cloud = O.occupancy(pts, O.occ_params(0.0, 4.5, 0.0, 3.78, -0.5, 0.6, 0.15, 0.5))["xyz"]
vox, _ = O.voxel_downsample(cloud, 0.02)
c = capi.Context(len(vox), profiling=True)
for _ in range(3):
    got = c.statistical_outlier_removal(vox, 15, 1.0)
print({nm: round(1e3 * ms, 1) for nm, ms in c.stage_times()})
nw = (len(vox) + 63) // 64
st = np.array(c.debug_buffer("idx_b", np.uint32, 4 * nw)).reshape(nw, 4).astype(np.int64)
for k, nm in enumerate(["27 cells", "shells", "blocks"]):
    v = st[:, k] / 100.0
    print("%-9s us per wave: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (nm, v.mean(), *np.percentile(v, [50, 90, 99]), v.max()))
u = st[:, 3]
print("waves", nw, "with lanes going on:", (u > 0).sum(), "lanes going on per wave: mean %.1f max %d" % (u.mean(), u.max()))
tot = st[:, :3].sum(1) / 100.0
print("whole wave: mean %.1f p99 %.1f max %.1f us;  sum over waves / 2048 slots = %.1f us" % (tot.mean(), np.percentile(tot, 99), tot.max(), tot.sum() / 2048))
o = np.argsort(-tot)[:5]
print("slowest waves:", [(int(i), st[i].tolist()) for i in o])
c.close()
