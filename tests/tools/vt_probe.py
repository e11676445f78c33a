"""Does the tiled voxel path engage on full-size C3 frames?  Prints DevState fields per round."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pointcloud_obstacle_processing_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
p = synth.params_of("C3")
c = capi.Context(n, max_frames=B, profiling=True)
c.set_params(publish_clouds=0, **p)
ptrs = [c.upload_soa(synth.make_cloud("C3", n=n, seed=3000 + i)) for i in range(B)]
for r in range(5):
    c.submit_batch_device([q for q, _ in ptrs], [m for _, m in ptrs])
    out = c.wait_batch(want_arrays=False)
    st = c.debug_state()
    print("round", r, "vt", st.vt, "err", st.err, "nb", st.nbuckets, "nb_cap", st.nb_cap, "gbits", st.gbits, "cshift", st.cshift, "maxbin", st.maxbin,
          "min_b", list(st.min_b), "div_b", list(st.div_b), "m", st.m, "k", out[0]["num_clusters"])
    print("   ", [(nm, round(ms * 1e3, 1)) for nm, ms in c.stage_times()])
