"""The C3 cloud in different point orders (a sensor delivers scan order, not a shuffle): batch throughput and single-frame latency,
every result compared with the oracle.  usage: order_bench.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth

p = synth.params_of("C3")
base = synth.make_cloud("C3")
inv = np.float32(1.0) / np.float32(p["leaf"])
ijk = np.floor(base * inv).astype(np.int64)
orders = {
    "as generated (shuffled)": np.arange(len(base)),
    "sorted along x": np.argsort(base[:, 0], kind="stable"),
    "sorted by voxel key (z, y, x)": np.lexsort((ijk[:, 0], ijk[:, 1], ijk[:, 2])),
    "scan order (elevation rows, azimuth inside a row, seen from the origin)": np.lexsort((np.arctan2(base[:, 1], base[:, 0] + 1e-3), np.round(np.arctan2(base[:, 2], np.hypot(base[:, 0], base[:, 1]) + 1e-3) * 200))),
}
B = 8
for name, perm in orders.items():
    pts = np.ascontiguousarray(base[perm])
    ctx = capi.Context(len(pts), max_frames=B); ctx.set_params(publish_clouds=1, **p)
    got = ctx.process_xyz(pts)
    ref = O.process_xyz(pts, **p)
    ok = got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes() and np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])
    ctx.set_params(publish_clouds=0, **p)
    ptr, _ = ctx.upload_soa(pts)
    for _ in range(3): ctx.process_device(ptr, len(pts))
    lat = []
    for _ in range(10):
        t = time.perf_counter(); ctx.submit_device(ptr, len(pts)); ctx.wait(want_arrays=False); lat.append(time.perf_counter() - t)
    for _ in range(3):
        ctx.submit_batch_device([ptr] * B, [len(pts)] * B); ctx.wait_batch(want_arrays=False)
    t = time.perf_counter()
    for _ in range(10):
        ctx.submit_batch_device([ptr] * B, [len(pts)] * B); ctx.wait_batch(want_arrays=False)
    fps = 10 * B / (time.perf_counter() - t)
    st = ctx.debug_state()
    print("%-75s bit-exact %s | latency %.3f ms | %8.1f frames/s (batch %d, 1 ctx) | voxel path %s, replans %d" %
          (name, ok, 1e3 * float(np.median(lat)), fps, B, "stream" if st.vpath else "sort", ctx.debug_counter("replans")), flush=True)
    ctx.close()
