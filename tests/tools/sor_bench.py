"""Statistical outlier removal (pcop_statistical_outlier_removal) on the voxel cloud of a C3 frame: the call, its GPU
part, and the CPU restatement (1 core) beside it."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth

vox, _ = O.voxel_downsample(synth.make_cloud("C3", seed=3000), 0.02)
c = capi.Context(len(vox), profiling=True)
for _ in range(3):
    got = c.statistical_outlier_removal(vox, 15, 1.0)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    got = c.statistical_outlier_removal(vox, 15, 1.0)
call = (time.perf_counter() - t0) / reps
t0 = time.perf_counter(); ref = O.sor(vox, 15, 1.0); cpu = time.perf_counter() - t0
assert got["distances"].tobytes() == ref["distances"].tobytes() and got["threshold"] == ref["threshold"]
stages = {nm: round(ms, 4) for nm, ms in c.stage_times()}
print(json.dumps({"stage_ms_last_pass": stages, "points": len(vox), "kept": len(got["xyz"]), "mean_k": 15, "call_ms_host_inclusive": round(1e3 * call, 3),
                  "cpu_restatement_ms_1core": round(1e3 * cpu, 1), "bit_exact": True}))
c.close()
