"""Frames/s of one configuration the way the bench line is measured - NC contexts, a feeder thread each, B device-resident frames a
batch - for the configurations the bench does not time (tests/tools/config_report.py times ONE context).  One synthetic cloud,
uploaded once per slot (every slot its own device buffer).
usage: pipelined_rate.py CFG [contexts=3] [batch=2] [seconds=3]      e.g.  pipelined_rate.py C5 3 2"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa  (its HIP runtime first, as in bench.py)
from pointcloud_obstacle_processing_amd import capi, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
p = synth.params_of(cfg); pts = synth.make_cloud(cfg); n = len(pts)
rows = []
for nc in sorted({1, NC}):
    ctxs = []
    for _ in range(nc):
        c = capi.Context(n, max_frames=B); c.set_params(publish_clouds=0, **p)
        up = [c.upload_soa(pts) for _ in range(B)]
        ctxs.append((c, [q for q, _ in up], [m for _, m in up]))
    for c, pp, nn in ctxs:   # settle the plans (bounds pass, then the box)
        for _ in range(3):
            c.submit_batch_device(pp, nn); c.wait_batch(want_arrays=False)
    done = [0] * nc
    stop = time.perf_counter() + secs

    def feed(i):
        c, pp, nn = ctxs[i]
        while time.perf_counter() < stop:
            c.submit_batch_device(pp, nn); c.wait_batch(want_arrays=False); done[i] += B

    t0 = time.perf_counter()
    th = [threading.Thread(target=feed, args=(i,)) for i in range(nc)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    st = ctxs[0][0].debug_state()
    rows.append("%d ctx x batch %d: %.1f frames/s" % (nc, B, sum(done) / dt))
    for c, _, _ in ctxs: c.close()
print("%s N=%d M=%d cells=%d: %s" % (cfg, n, st.m, st.n_cells, "; ".join(rows)), flush=True)
