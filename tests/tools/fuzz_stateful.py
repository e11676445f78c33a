"""One long-lived context fed a random sequence of clouds whose sizes and parameters swing widely, so that every adaptive
piece of host state moves (radix pass counts, LDS per bucket, scan grids, voxel / cell path fallbacks and their way back,
crowded-cell switch); every frame is compared with the CPU oracle.  usage: fuzz_stateful.py [steps] [seed]"""
import os, sys, time
import numpy as np
os.environ.setdefault("PCOP_VOXEL_PATH", "stream")   # the adaptive state worth shaking is the streaming path's (single frames would take the radix path by default)
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
    ("C2", dict(leaf=0.2, tol=0.01), 30.0),            # the voxel grid streams, the cell grid is far too large: streaming path + hash table of cells
    ("C1", dict(leaf=0.0005, tol=0.05), 1.0),          # voxel grid beyond 2^26 cells -> sort path
    ("C3", dict(leaf=0.02, tol=0.1), 1.0),
    ("SHEET", dict(leaf=0.01, tol=0.05), 1.0),         # a dense sheet in a tall box: coarse bins above a bucket -> k_heavy_split
]


def sheet(n, seed):
    r = np.random.default_rng(seed)
    a = np.concatenate([r.uniform(0, 2, (n, 2)), r.normal(0.5, 0.001, (n, 1))], axis=1)
    b = r.uniform(0, 2, (max(n // 200, 8), 3))
    q = np.concatenate([a, b]).astype(f32)
    return q[r.permutation(len(q))][:n]


fails = 0
dirty = set()


def check_clean(it, tag, slots=(0,)):
    """FUZZ_CHECK_CLEAN=1: after every call the occupancy flags are zero and the cell table is empty, in every slot the call used."""
    global fails
    if not os.environ.get("FUZZ_CHECK_CLEAN"):
        return
    for sl in slots:
        for name, dt, want in (("cflag", np.uint8, 0), ("hcount", np.uint32, 0), ("hmin", np.uint32, 0xFFFFFFFF), ("hkey", np.uint64, 0)):
            nm = name if sl == 0 else "%s@%d" % (name, sl)
            bad = np.flatnonzero(ctx.debug_buffer(nm, dt) != want)
            if len(bad) and nm not in dirty:
                dirty.add(nm); fails += 1
                print("DIRTY after step %d %s: %s has %d entries off (first at %d, last at %d)" % (it, tag, nm, len(bad), bad[0], bad[-1]), flush=True)
                sys.exit(2)   # (a later frame would read cells without points: stop here)


t0 = time.time()
paths = set()
boxed = 0
for it in range(steps):
    cfg, prm, stretch = menu[int(rng.integers(0, len(menu)))]
    n = int(rng.choice([5_000, 150_000, 600_000, 1_000_000] if big else [5_000, 40_000, 150_000, 400_000]))
    seed = int(rng.integers(0, 1 << 30))
    pts = sheet(n, seed) if cfg == "SHEET" else (synth.make_cloud(cfg, n=n, seed=seed) * f32(stretch)).astype(f32)
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
    check_clean(it, "frame")
    if not ok:
        fails += 1
        print("MISMATCH step", it, cfg, n, stretch, p, flush=True)
    if it % 3 == 2:   # the same cloud and a smaller one as ONE BATCH on the same context (the box of the streaming path, the batch-only kernels)
        other = pts[: max(1, n // 3)][::-1].copy()
        try:
            ctx.submit_batch_host([pts, other])
            rb = ctx.wait_batch()
            st = ctx.debug_state()
            paths.add((st.vpath, st.cpath))
            boxed = boxed + (1 if st.sbox else 0)
            okb = True
            for fr, srcp in zip(rb, (pts, other)):
                rr = ref if srcp is pts else O.process_xyz(srcp, **p)
                okb = okb and fr["voxel_xyz"].tobytes() == rr["voxel_xyz"].tobytes() and np.array_equal(fr["offsets"], rr["offsets"]) and np.array_equal(fr["indices"], rr["indices"])
        except capi.PcopError as e:
            okb = False
            print("ERROR (batch)", e, flush=True)
        check_clean(it, "batch", (0, 1))
        if not okb:
            fails += 1
            print("BATCH MISMATCH step", it, cfg, n, stretch, p, flush=True)
    if it % 4 == 1 and n <= 150_000 and stretch == 1.0 and os.environ.get("PCOP_CELL_PATH") != "hash":   # the node's chain on the same context, in between
        mk = int(rng.choice([0, 15])); use_plane = bool(rng.integers(0, 2)); thr = 0.04
        try:
            cur, _ = O.voxel_downsample(pts, p["leaf"])
            if mk and len(cur) > mk: cur = O.sor(cur, mk, 1.0)["xyz"]
            elif mk: mk = 0
            if use_plane: cur = O.segment_planes(cur, thr)["xyz"]
            ro, ri = O.euclidean_clusters(cur, p["tol"], p["min_cluster"], p["max_cluster"])
            gn = ctx.process_node(pts, mean_k=mk, stddev_mul=1.0, plane=capi.plane_params(thr) if use_plane else None)
            gv = gn["voxel_xyz"] if gn["voxel_xyz"] is not None else np.zeros((0, 3), dtype=f32)
            okn = gv.tobytes() == cur.tobytes() and np.array_equal(gn["offsets"], ro) and np.array_equal(gn["indices"], ri)
        except capi.PcopError as e:
            okn = False
            print("ERROR (node chain)", e, flush=True)
        check_clean(it, "node chain")
        if not okn:
            fails += 1
            print("NODE CHAIN MISMATCH step", it, cfg, n, p, mk, use_plane, flush=True)
    if it % 10 == 9:
        print("  ... %d steps, %d mismatches, %.0f s, (voxel, cell) paths seen %s" % (it + 1, fails, time.time() - t0, sorted(paths)), flush=True)
print("steps %d, mismatches %d, (voxel, cell) paths %s, batches keyed on the context's box %d, replans %d, %.1f s" %
      (steps, fails, sorted(paths), boxed, ctx.debug_counter("replans"), time.time() - t0))
ctx.close()
sys.exit(1 if fails else 0)
