"""One long-lived context fed a random sequence of clouds whose sizes and parameters swing widely, so that every adaptive
piece of host state moves (radix pass counts, LDS per bucket, scan grids, voxel / cell path fallbacks and their way back,
crowded-cell switch); every frame is compared with the CPU oracle.  usage: fuzz_stateful.py [steps] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth

f32 = np.float32
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = os.environ.get("FUZZ_BIG") == "1"     # FUZZ_BIG=1: clouds up to 1 M points (slower oracle)
ctx = capi.Context(1_000_000 if big else 400_000, max_frames=2)
menu = [
    ("C3", dict(leaf=0.02, tol=0.1), 1.0),            # the benchmark shape
    ("C3", dict(leaf=0.004, tol=0.02), 1.0),           # many voxels: more LDS per bucket; many cells
    ("C3", dict(leaf=0.05, tol=0.4), 1.0),             # crowded cells
    ("C2", dict(leaf=0.05, tol=0.2), 1.0),
    ("C2", dict(leaf=0.05, tol=0.01), 30.0),           # huge extent / tiny tolerance: cell grid too large -> hash table
    ("C1", dict(leaf=0.0005, tol=0.05), 1.0),          # voxel grid beyond 2^26 cells -> sort path
    ("C3", dict(leaf=0.02, tol=0.1), 1.0),
]
fails = 0
t0 = time.time()
paths = set()
for it in range(steps):
    cfg, prm, stretch = menu[int(rng.integers(0, len(menu)))]
    n = int(rng.choice([5_000, 150_000, 600_000, 1_000_000] if big else [5_000, 40_000, 150_000, 400_000]))
    pts = (synth.make_cloud(cfg, n=n, seed=int(rng.integers(0, 1 << 30))) * f32(stretch)).astype(f32)
    p = dict(prm, min_cluster=int(rng.choice([1, 5])), max_cluster=int(rng.choice([20000, 10**9])))
    ctx.set_params(publish_clouds=1, **p)
    try:
        got = ctx.process_xyz(pts)
        st = ctx.debug_state()
        paths.add((st.vpath, st.cpath))
        ref = O.process_xyz(pts, **p)
        ok = (got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes() and np.array_equal(got["offsets"], ref["offsets"])
              and np.array_equal(got["indices"], ref["indices"]))
    except capi.PcopError as e:
        ok = False
        print("ERROR", e, flush=True)
    if not ok:
        fails += 1
        print("MISMATCH step", it, cfg, n, stretch, p, flush=True)
    if it % 10 == 9:
        print("  ... %d steps, %d mismatches, %.0f s, (voxel, cell) paths seen %s" % (it + 1, fails, time.time() - t0, sorted(paths)), flush=True)
print("steps %d, mismatches %d, paths %s, %.1f s" % (steps, fails, sorted(paths), time.time() - t0))
ctx.close()
sys.exit(1 if fails else 0)
