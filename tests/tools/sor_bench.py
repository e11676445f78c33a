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
import ctypes as C
c = capi.Context(len(vox), profiling=True)
for _ in range(3):
    got = c.statistical_outlier_removal(vox, 15, 1.0)
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    got = c.statistical_outlier_removal(vox, 15, 1.0)
call_py = (time.perf_counter() - t0) / reps
# the C call alone (what a C++ node pays): buffers allocated once, no per-stage events on the stream
c2 = capi.Context(len(vox))
flat, n = capi.pack_points(vox, capi.LAYOUT_AOS12)
out, dist = np.zeros_like(flat), np.zeros(n, dtype=np.float32)
n_out, thr = C.c_size_t(0), C.c_double(0.0)
def one():
    rc = c2.lib.pcop_statistical_outlier_removal(c2.h, flat.ctypes.data, capi.LAYOUT_AOS12, n, 15, C.c_float(1.0), out.ctypes.data,
                                                 C.byref(n_out), dist.ctypes.data, C.byref(thr))
    assert rc == 0
for _ in range(3): one()
t0 = time.perf_counter()
for _ in range(reps): one()
call = (time.perf_counter() - t0) / reps
assert dist.tobytes() == got["distances"].tobytes() and thr.value == got["threshold"] and n_out.value == len(got["xyz"])
c2.close()
t0 = time.perf_counter(); ref = O.sor(vox, 15, 1.0); cpu = time.perf_counter() - t0
assert got["distances"].tobytes() == ref["distances"].tobytes() and got["threshold"] == ref["threshold"]
stages = {nm: round(ms, 4) for nm, ms in c.stage_times()}
print(json.dumps({"stage_ms_last_pass": stages, "points": len(vox), "kept": len(got["xyz"]), "mean_k": 15, "call_ms_host_inclusive": round(1e3 * call, 3), "call_ms_through_python_wrapper": round(1e3 * call_py, 3),
                  "cpu_restatement_ms_1core": round(1e3 * cpu, 1), "bit_exact": True}))
c.close()
