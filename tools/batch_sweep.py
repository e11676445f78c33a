"""Throughput vs batch size (frames per launch) and contexts in flight."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from pointcloud_obstacle_processing_amd import capi, synth
p = synth.params_of("C3")
clouds = [synth.make_cloud("C3", seed=3000 + i) for i in range(4)]
combos = [(1, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 3), (8, 4), (16, 2), (6, 3)]
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, NC in combos:
    ctxs = []
    for ci in range(NC):
        c = capi.Context(1_000_000, max_frames=B); c.set_params(publish_clouds=0, **p)
        up = [c.upload_soa(clouds[(ci * B + i) % len(clouds)]) for i in range(B)]
        ctxs.append((c, [u[0] for u in up], [u[1] for u in up]))
    for _ in range(3):
        for c, pp, nn in ctxs: c.submit_batch_device(pp, nn)
        for c, _, _ in ctxs: c.wait_batch(want_arrays=False)
    steps = max(4, 128 // (B * NC))
    t0 = time.perf_counter()
    for _ in range(steps):
        for c, pp, nn in ctxs: c.submit_batch_device(pp, nn)
        for c, _, _ in ctxs: c.wait_batch(want_arrays=False)
    dt = time.perf_counter() - t0
    # rolling: every context always has a batch queued; wait for the oldest, resubmit it at once
    for c, pp, nn in ctxs: c.submit_batch_device(pp, nn)
    t0 = time.perf_counter()
    for _ in range(steps):
        for c, pp, nn in ctxs:
            c.wait_batch(want_arrays=False); c.submit_batch_device(pp, nn)
    dtr = time.perf_counter() - t0
    for c, _, _ in ctxs: c.wait_batch(want_arrays=False)
    print("   rolling: %8.1f frames/s" % (steps * B * NC / dtr), flush=True)
    print("batch %2d x ctx %d: %8.1f frames/s  (%.1f us/frame, step %.2f ms)" % (B, NC, steps * B * NC / dt, 1e6 * dt / (steps * B * NC), 1e3 * dt / steps), flush=True)
    for c, _, _ in ctxs: c.close()
