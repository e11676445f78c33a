"""Batches of two to five frames on one long-lived context: frames of unlike shapes and sizes in one launch sequence, the
parameters changing only now and then (so that the context's voxel box, bucket plans and tile counts live for a while and then
have to move), frames that fall outside the box, frames with NaN holes, empty and one-point frames.  Every frame is compared with
the CPU oracle; FUZZ_CHECK_CLEAN=1 checks every slot's occupancy flags and cell table after every batch.
usage: fuzz_batches.py [steps] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi, synth

f32 = np.float32
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CAP, NF = 120_000, 5
ctx = capi.Context(CAP, max_frames=NF)


def frame(scale, centre):
    kind = rng.choice(["c3", "cube", "plane", "line", "blobs", "tiny", "empty"], p=[0.3, 0.15, 0.15, 0.1, 0.15, 0.1, 0.05])
    n = int(rng.choice([5_000, 20_000, 60_000, 120_000]))
    if kind == "c3":
        p = synth.make_cloud("C3", n=n, seed=int(rng.integers(0, 1 << 30))) * f32(scale / 4.5)
    elif kind == "cube":
        p = rng.uniform(-scale, scale, (n, 3))
    elif kind == "plane":
        p = np.concatenate([rng.uniform(-scale, scale, (n, 2)), rng.normal(0, 0.002, (n, 1))], axis=1)
    elif kind == "line":
        n = min(n, 20_000); p = np.zeros((n, 3)); p[:, 0] = rng.uniform(-scale, scale, n)
    elif kind == "blobs":
        c = rng.uniform(-scale, scale, (max(1, n // 500), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.03, (n, 3))
    elif kind == "tiny":
        p = rng.uniform(-scale, scale, (int(rng.choice([1, 2, 3, 7, 65])), 3))
    else:
        p = np.zeros((0, 3))
    return (p + centre).astype(f32)


def check_clean(it, nf):
    if not os.environ.get("FUZZ_CHECK_CLEAN"):
        return
    for sl in range(nf):
        for name, dt, want in (("cflag", np.uint8, 0), ("hcount", np.uint32, 0), ("hmin", np.uint32, 0xFFFFFFFF), ("hkey", np.uint64, 0)):
            nm = name if sl == 0 else "%s@%d" % (name, sl)
            bad = np.flatnonzero(ctx.debug_buffer(nm, dt) != want)
            if len(bad):
                print("DIRTY after step %d: %s has %d entries off (first at %d, last at %d)" % (it, nm, len(bad), bad[0], bad[-1]), flush=True)
                sys.exit(2)


fails = 0
t0 = time.time()
p = None
for it in range(steps):
    if p is None or rng.random() < 0.15:     # new parameters now and then
        leaf = float(rng.choice([0.01, 0.02, 0.05]))
        p = dict(leaf=leaf, tol=float(rng.choice([2.0, 5.0])) * leaf, min_cluster=int(rng.choice([1, 5])), max_cluster=int(rng.choice([20000, 10**9])))
        scale = float(rng.choice([0.5, 2.0, 4.5]))
        ctx.set_params(publish_clouds=1, **p)
    nf = int(rng.integers(2, NF + 1))
    clouds = []
    for f in range(nf):
        far = rng.random() < 0.1             # a frame elsewhere: outside the box the context has learnt
        centre = rng.uniform(-3 * scale, 3 * scale, 3) if far else np.zeros(3)
        clouds.append(frame(scale, centre))
    dense = True
    if rng.random() < 0.2:                   # NaN holes, the cloud declared not dense
        dense = False
        for c in clouds:
            if len(c) > 10:
                c[::13] = np.nan
    refs = []
    try:
        for c in clouds:
            refs.append(O.process_xyz(c, is_dense=dense, **p))
    except ValueError:
        refs = None                          # (the reference is undefined on this batch: the product must refuse it too)
    try:
        ctx.submit_batch_host(clouds, is_dense=dense)
        got = ctx.wait_batch()
        ok = refs is not None
        for g, r in zip(got, refs or []):
            gv = g["voxel_xyz"] if g["voxel_xyz"] is not None else np.zeros((0, 3), dtype=f32)   # (a frame without points has no cloud)
            ok = ok and gv.tobytes() == r["voxel_xyz"].tobytes() and np.array_equal(g["offsets"], r["offsets"]) and np.array_equal(g["indices"], r["indices"])
    except capi.PcopError as e:
        ok = refs is None
        if not ok:
            print("ERROR", e, flush=True)
    check_clean(it, nf)
    if not ok:
        fails += 1
        print("MISMATCH step", it, p, [len(c) for c in clouds], dense, flush=True)
    if it % 20 == 19:
        print("  ... %d steps, %d mismatches, %.0f s, replans %d" % (it + 1, fails, time.time() - t0, ctx.debug_counter("replans")), flush=True)
print("steps %d, mismatches %d, replans %d, re-runs after a bounded wait %d, %.1f s" % (steps, fails, ctx.debug_counter("replans"), ctx.debug_counter("internal_retries"), time.time() - t0))
ctx.close()
sys.exit(1 if fails else 0)
