"""Context life cycle: create (1 or 8 frame slots), run frames (chain, stage calls, occupancy, outlier removal), destroy -
repeatedly in one process.  Looks for anything the runtime objects to at teardown."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from pointcloud_obstacle_processing_amd import capi, synth

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
p = synth.params_of("C3")
pts = synth.make_cloud("C3", n=150_000, seed=1)
for i in range(cycles):
    c = capi.Context(1_100_000 if i % 3 == 0 else 200_000, max_frames=8 if i % 2 else 1)
    c.set_params(publish_clouds=1, **p)
    r = c.process_xyz(pts)
    if i % 2:
        ptr, n = c.upload_soa(pts)
        c.submit_batch_device([ptr] * 8, [n] * 8)
        c.wait_batch(want_arrays=False)
    c.voxel_downsample(pts)
    c.statistical_outlier_removal(r["voxel_xyz"], 15, 1.0)
    c.close()
    print("cycle", i, "ok", r["num_voxels"], flush=True)
print("done")
