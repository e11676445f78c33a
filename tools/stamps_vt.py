"""Diagnostic build only (make HIPFLAGS+=-DPCOP_STAMPS): cycles between the phase stamps of k_vt_sort and k_vt_reduce."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloud_obstacle_processing_amd import capi, synth
B = 16
p = synth.params_of("C3")
c = capi.Context(1_000_000, max_frames=B); c.set_params(publish_clouds=0, **p)
ptrs = [c.upload_soa(synth.make_cloud("C3", seed=3000 + i)) for i in range(B)]
for rep in range(4):
    c.submit_batch_device([q for q, _ in ptrs], [m for _, m in ptrs]); c.wait_batch(want_arrays=False)
print("vt", c.debug_state().vt)
st = np.array(c.debug_buffer("hsub", np.uint64, 64)).reshape(2, 32).astype(np.int64)
for k, nm in enumerate(["k_vt_sort (block 40)", "k_vt_reduce (block 7, its last bucket)"]):
    row = st[k]
    print("   k_vt_accum, block 7 wave 0, cycles over all its buckets: table %d, clear %d, pass(direct) %d, emit %d, ranked buckets %d, loop top %d" % tuple(row[24:30]))
    idx = sorted([i for i in range(20) if row[i] and abs(int(row[i]) - int(row[0])) < 10**8], key=lambda i: row[i])
    print(nm)
    prev = None
    for i in idx:
        print("   stamp %2d  +%8d cycles" % (i, 0 if prev is None else row[i] - prev))
        prev = row[i]
c.close()
