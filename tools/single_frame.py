"""One C3 frame at a time through the frame API (the node's mode of use: queue depth 1, od.cpp:1001), for a kernel trace."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from pointcloud_obstacle_processing_amd import capi, synth
p = synth.params_of("C3")
c = capi.Context(1_000_000); c.set_params(publish_clouds=0, **p)
ptr, n = c.upload_soa(synth.make_cloud("C3", seed=3000))
for _ in range(5):
    c.process_device(ptr, n)
lat = []
for _ in range(30):
    t0 = time.perf_counter(); c.submit_device(ptr, n); c.wait(want_arrays=False); lat.append(time.perf_counter() - t0)
print("single-frame latency: median %.4f ms" % (1e3 * float(np.median(lat))))
c.close()
