"""Single-frame latency under two or more settings of one environment switch, interleaved in one run on one box.
usage: latency_env.py VAR "v1 v2 ..." [C3 C3L10 ...]      e.g.  latency_env.py PCOP_HOOK2 "0 1" C3 C3L10"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from pointcloud_obstacle_processing_amd import capi, synth
var, vals = sys.argv[1], sys.argv[2].split()
for cfg in (sys.argv[3:] or ["C3"]):
    p = synth.params_of(cfg); pts = synth.make_cloud(cfg)
    row = []
    iters = 40 if len(pts) > 2_000_000 else 120
    for rep in range(2):
        for v in vals:
            if v == "-":
                os.environ.pop(var, None)
            else:
                os.environ[var] = v
            c = capi.Context(len(pts)); c.set_params(publish_clouds=0, **p)
            ptr, n = c.upload_soa(pts)
            lat = []
            for it in range(iters):
                t = time.perf_counter(); c.submit_batch_device([ptr], [n]); c.wait_batch(want_arrays=False); lat.append(time.perf_counter() - t)
            st = c.debug_state()
            row.append("%s=%s %.4f ms" % (var, v, 1e3 * float(np.median(lat[iters // 5:]))))
            cells, m = st.n_cells, st.c_n
            c.close()
    print("%s: %d cells, %.1f points a cell: %s" % (cfg, cells, m / max(1, cells), ", ".join(row)), flush=True)
