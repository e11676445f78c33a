"""After k_cgrid_hook + k_cell_compress (PCOP_DEBUG_SKIP_UNION=1): how many trees, and where, on a field frame with strays."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PCOP_DEBUG_SKIP_UNION"] = "1"
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth
from test_gpu_node_chain import field_cloud
pts = field_cloud(1_000_000, 77)
cloud = O.occupancy(pts, O.occ_params(0.0, 4.5, 0.0, 3.78, -0.5, 0.6, 0.15, 0.5))["xyz"]
for name, vox in (("field+strays", O.voxel_downsample(cloud, 0.02)[0]), ("C3", O.voxel_downsample(synth.make_cloud("C3", seed=3000), 0.02)[0])):
    c = capi.Context(len(vox)); c.set_params(leaf=0.02, tol=0.1, min_cluster=5, max_cluster=20000, publish_clouds=0)
    c.euclidean_clusters(vox); st = c.debug_state()
    nc = st.n_cells
    parent = c.debug_buffer("parent", np.uint32, nc)
    meta = c.debug_buffer("slot_meta", np.uint32, 4 * nc).reshape(nc, 4)
    pos = meta[:, 3].astype(np.int64); cnt = meta[:, 1]
    gz = pos // st.c_plane; gy = (pos % st.c_plane) // st.c_rowbits; gx = pos % st.c_rowbits
    isroot = parent == np.arange(nc)
    print(name, "cells", nc, "roots", isroot.sum(), "distinct parent words", len(np.unique(parent)))
    zs, n_all = np.unique(gz, return_counts=True)
    for z, n in zip(zs, n_all):
        m = gz == z
        if n > 200: print("   z layer %d: cells %d roots %d mean pts/cell %.1f" % (z, n, isroot[m].sum(), cnt[m].mean()))
    # tree sizes
    _, sizes = np.unique(parent, return_counts=True)
    print("   tree sizes: max %d, >100 cells: %d, singletons: %d" % (sizes.max(), (sizes > 100).sum(), (sizes == 1).sum()))
    c.close()
