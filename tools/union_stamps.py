"""Diagnostic build only (make HIPFLAGS+=-DPCOP_STAMPS): per wave of k_cgrid_union, the time of its two parts and the work of
the slow one, on the voxel cloud of a field frame with strays (cluster-only call) and on a C3 voxel cloud."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth
from test_gpu_node_chain import field_cloud
os.environ["PCOP_GRAPH"] = "0"
def run(name, vox, tol):
    c = capi.Context(len(vox), profiling=True)
    c.set_params(leaf=0.02, tol=tol, min_cluster=5, max_cluster=20000, publish_clouds=0)
    for _ in range(3):
        off, idx = c.euclidean_clusters(vox)
    st = c.debug_state()
    print(name, "points", len(vox), "cells", st.n_cells, "clusters", len(off) - 1, {nm: round(1e3 * ms, 1) for nm, ms in c.stage_times() if "union" in nm or "hook" in nm or "compress" in nm})
    nw = (st.n_cells + 63) // 64
    d = np.array(c.debug_buffer("rec_b", np.uint32, 8 * nw)).reshape(nw, 8).astype(np.int64)
    for k, nm in enumerate(["fast part", "slow part"]):
        v = d[:, k] / 100.0
        print("  %-9s us per wave: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (nm, v.mean(), *np.percentile(v, [50, 90, 99]), v.max()))
    print("  waves", nw, "with slow lanes", (d[:, 2] > 0).sum(), "slow lanes total", d[:, 2].sum(), "pair tests", d[:, 3].sum(), "unions", d[:, 4].sum(), "marked slots", d[:, 5].sum(), "roots before union", d[:, 6].sum())
    o = np.argsort(-d[:, 1])[:5]
    print("  slowest:", [(int(i), d[i, :7].tolist()) for i in o])
    c.close()
pts = field_cloud(1_000_000, 77)
cloud = O.occupancy(pts, O.occ_params(0.0, 4.5, 0.0, 3.78, -0.5, 0.6, 0.15, 0.5))["xyz"]
vox, _ = O.voxel_downsample(cloud, 0.02)
run("field+strays", vox, 0.1)
vox3, _ = O.voxel_downsample(synth.make_cloud("C3", seed=3000), 0.02)
run("C3", vox3, 0.1)
