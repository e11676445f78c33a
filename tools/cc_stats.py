import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PCOP_DEBUG_SKIP_UNION"] = "1"
from pointcloud_obstacle_processing_amd import capi, synth
p = synth.params_of("C3"); pts = synth.make_cloud("C3", seed=3000)
c = capi.Context(len(pts)); c.set_params(publish_clouds=0, **p)
r = c.process_xyz(pts); st = c.debug_state()
cap = st.hash_mask + 1
cell_list = c.debug_buffer("cell_list", np.uint32, st.n_cells)
cell_start = c.debug_buffer("cell_start", np.uint32, cap)
hmin_dummy = None
rec = c.debug_buffer("rec_a", np.float32, 4 * st.c_n).reshape(-1, 4)
parent = c.debug_buffer("parent", np.uint32, cap)
ig = st.inv_cell
first = rec[cell_start[cell_list]][:, :3].astype(np.float64)
cc = np.floor(first * ig).astype(int)
roots = parent[cell_list]
isroot = roots == cell_list
print("cells", st.n_cells, "trees after hook+compress:", isroot.sum(), " distinct parents", len(np.unique(roots)))
rz = cc[isroot]
print("root cells by z:", dict(zip(*np.unique(rz[:, 2], return_counts=True))))
# look at one box: roots with z>=3, print a few root coordinates and their backward neighbours' presence
occ = {tuple(v): i for i, v in enumerate(cc)}
prio = rec[cell_start[cell_list]][:, 3].view(np.uint32)   # first record idx ~ NOT min idx (unordered in cell); approximate
shown = 0
for i in np.flatnonzero(isroot & (cc[:, 2] >= 3))[:8]:
    x, y, z = cc[i]
    nb = []
    for dz in (-1, 0):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if (dz, dy, dx) >= (0, 0, 0): continue
                j = occ.get((x + dx, y + dy, z + dz))
                if j is not None: nb.append(((dx, dy, dz), int(cell_start[cell_list[j]]) ))
    print("root", (x, y, z), "backward occupied neighbours:", [n[0] for n in nb])
