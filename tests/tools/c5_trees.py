"""C5 (5 M points, one giant component) after k_cgrid_hook + k_cell_compress (PCOP_DEBUG_SKIP_UNION=1): how many trees the
union kernel has to merge, and how large they are.  With an argument `hook2`: after the second, tree-level hook round as well
(PCOP_HOOK2=1, PCOP_DEBUG_SKIP_UNION=2: the union kernel scans but never unites)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HOOK2 = len(sys.argv) > 1 and sys.argv[1] == "hook2"
os.environ["PCOP_DEBUG_SKIP_UNION"] = "2" if HOOK2 else "1"
os.environ["PCOP_HOOK2"] = "1" if HOOK2 else "0"
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth
p = synth.params_of("C5")
vox = O.voxel_downsample(synth.make_cloud("C5"), p["leaf"])[0]
c = capi.Context(len(vox)); c.set_params(publish_clouds=0, **p)
c.euclidean_clusters(vox); st = c.debug_state()
nc = st.n_cells
parent = c.debug_buffer("parent", np.uint32, nc)
meta = c.debug_buffer("slot_meta", np.uint32, 4 * nc).reshape(nc, 4)
roots = int((parent == np.arange(nc)).sum())
_, sizes = np.unique(parent, return_counts=True)
print(("C5: voxels %d, cells %d (%.2f points a cell), trees after hook + compress" + (" + hook2 + compress2" if HOOK2 else "") + " %d (%.1f cells a tree; largest %d, singletons %d)")
      % (len(vox), nc, meta[:, 1].mean(), roots, nc / max(roots, 1), sizes.max(), int((sizes == 1).sum())))
c.close()
