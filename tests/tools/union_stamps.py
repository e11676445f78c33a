"""The cluster stage's union on a clean voxel cloud and on one with strays: stage times, cells k_cgrid_union hands to
k_cgrid_fix."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth
from test_gpu_node_chain import field_cloud
def run(name, vox, tol):
    c = capi.Context(len(vox), profiling=True)
    c.set_params(leaf=0.02, tol=tol, min_cluster=5, max_cluster=20000, publish_clouds=0)
    for _ in range(3):
        off, idx = c.euclidean_clusters(vox)
    st = c.debug_state()
    print(name, "points", len(vox), "cells", st.n_cells, "clusters", len(off) - 1,
          {nm: round(1e3 * ms, 1) for nm, ms in c.stage_times() if "union" in nm or "hook" in nm or "compress" in nm or "fix" in nm})
    ro, ri = O.euclidean_clusters(vox, tol, 5, 20000)
    print("   bit-exact:", np.array_equal(off, ro) and np.array_equal(idx, ri))
    c.close()
pts = field_cloud(1_000_000, 77)
cloud = O.occupancy(pts, O.occ_params(0.0, 4.5, 0.0, 3.78, -0.5, 0.6, 0.15, 0.5))["xyz"]
run("field+strays", O.voxel_downsample(cloud, 0.02)[0], 0.1)
run("C3", O.voxel_downsample(synth.make_cloud("C3", seed=3000), 0.02)[0], 0.1)
