import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth
from test_gpu_node_chain import field_cloud, occ_of
pts = field_cloud(1_000_000, 77)
po, pc = occ_of(capi, O)
c = capi.Context(len(pts), profiling=True)
c.set_params(leaf=0.02, tol=0.1, min_cluster=5, max_cluster=20000, publish_clouds=1)
dev, _ = c.upload_soa(pts)
for kw in (dict(occ=pc), dict(occ=pc, mean_k=15, stddev_mul=1.0), dict(occ=pc, mean_k=15, stddev_mul=1.0, plane=capi.plane_params(0.04))):
    for _ in range(4):
        t0 = time.perf_counter(); got = c.process_node(None, dev_ptr=dev, n=len(pts), is_dense=False, **kw); t = time.perf_counter() - t0
    st = c.stage_times()
    print("call %.3f ms; gpu stages sum %.3f ms" % (1e3 * t, sum(ms for _, ms in st)))
    print("  " + ", ".join("%s %.1f" % (nm, 1e3 * ms) for nm, ms in st))
