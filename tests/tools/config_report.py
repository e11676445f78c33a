"""Every BASELINE config through the GPU path and the CPU oracle: parity + timings (for BASELINE.md §5)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth

cfgs = sys.argv[1:] or ["C1", "C2", "C3", "C3L10", "C5", "YAML"]
for cfg in cfgs:
    if cfg == "YAML":   # the reference's deployed parameters (params.yaml:16,26-28) on the C3 cloud
        p = dict(leaf=0.015, tol=0.4, min_cluster=5, max_cluster=20000); pts = synth.make_cloud("C3")
    else:
        p = synth.params_of(cfg); pts = synth.make_cloud(cfg)
    n = len(pts)
    B = 8 if n <= 1_000_000 else 2
    ctx = capi.Context(n, max_frames=B); ctx.set_params(publish_clouds=1, **p)
    got = ctx.process_xyz(pts)
    t = time.perf_counter(); ref = O.process_xyz(pts, **p); t_or = time.perf_counter() - t
    ok_vox = got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes()
    ok_idx = np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"])
    err = float(np.max(np.abs(got["voxel_xyz"] - ref["voxel_xyz"]))) if got["voxel_xyz"].shape == ref["voxel_xyz"].shape and len(ref["voxel_xyz"]) else -1.0
    # timings: single frame latency and batched throughput, cloud resident in HBM
    ctx.set_params(publish_clouds=0, **p)
    ptr, _ = ctx.upload_soa(pts)
    for _ in range(3): ctx.process_device(ptr, n)
    lat = []
    for _ in range(10):
        t = time.perf_counter(); ctx.submit_device(ptr, n); ctx.wait(want_arrays=False); lat.append(time.perf_counter() - t)
    for _ in range(2):
        ctx.submit_batch_device([ptr] * B, [n] * B); ctx.wait_batch(want_arrays=False)
    reps = 10
    t = time.perf_counter()
    for _ in range(reps):
        ctx.submit_batch_device([ptr] * B, [n] * B); ctx.wait_batch(want_arrays=False)
    fps = reps * B / (time.perf_counter() - t)
    st = ctx.debug_state()
    print("%-6s N=%8d M=%8d K=%4d cells=%7d | centroids bit-exact %s (max abs err %.1e) indices bit-exact %s | "
          "GPU latency %.3f ms, %8.1f frames/s (batch %d, 1 ctx) | oracle 1 core %.2f s = %.3f frames/s | voxel path %s, csr passes %d"
          % (cfg, n, got["num_voxels"], len(got["offsets"]) - 1, st.n_cells, ok_vox, err, ok_idx,
             1e3 * float(np.median(lat)), fps, B, t_or, 1.0 / t_or,
             "stream (%d buckets, %d coarse bins)" % (st.nbuckets, 1 << st.gbits) if st.vpath else "sort (%d passes)" % st.plan_voxel.npass,
             st.plan_csr.npass), flush=True)
    ctx.close()
