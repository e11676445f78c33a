"""Randomised parity campaign: many small and mid-sized clouds of awkward shapes through the GPU chain, the occupancy
step, the outlier removal, the plane removal and the node's chain in one call, each compared with the CPU oracle.  usage: python tests/tools/fuzz.py [cases] [seed]   (FUZZ_VERBOSE=1: every case and step; FUZZ_START=k: start at case k of the seed)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
from oracle import oracle as O
from pointcloud_obstacle_processing_amd import capi

f32 = np.float32
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


kinds = ["cube", "plane", "line", "blobs", "dups", "lattice"]


def cloud(kind, n, scale):
    if kind == "cube":
        p = rng.uniform(-scale, scale, (n, 3))
    elif kind == "plane":
        p = np.concatenate([rng.uniform(-scale, scale, (n, 2)), rng.normal(0, 0.002, (n, 1))], axis=1)
    elif kind == "line":
        p = np.zeros((n, 3)); p[:, 0] = rng.uniform(-scale, scale, n)
    elif kind == "blobs":
        c = rng.uniform(-scale, scale, (max(1, n // 200), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.03, (n, 3))
    elif kind == "dups":
        b = rng.uniform(-scale, scale, (max(1, n // 5), 3))
        p = b[rng.integers(0, len(b), n)]
    else:   # lattice: coordinates on multiples of a step, so that keys and distances sit on boundaries
        step = rng.choice([0.01, 0.02, 0.025, 0.05])
        p = rng.integers(-int(scale / step), int(scale / step) + 1, (n, 3)) * step
    return p.astype(f32)


def dry(it, kind, n, scale, pts):
    """FUZZ_START=k: the random draws of a case without its GPU / oracle calls (same draws, same order as the loop below),
    so that a run can start at case k of a seed with the generator where the full run would have it."""
    if it % 5 == 1 and n >= 3:
        rng.integers(0, 3)
    if it % 7 == 2:
        cloud(kinds[(it + 1) % len(kinds)], max(1, n // 3), scale)
    if it % 3 == 0 and n >= 33 and os.environ.get("PCOP_CELL_PATH") != "hash":
        rng.choice([1, 4, 15, 31])
    if it % 4 == 0:
        rng.choice([0.013, 0.05, 0.15])
    if it % 6 == 3 and n >= 3 and os.environ.get("PCOP_CELL_PATH") != "hash":
        rng.choice([0.005, 0.02, 0.1]); rng.choice([0, 4, 15]); rng.integers(0, 2)


start = int(os.environ.get("FUZZ_START", "0"))
ctx = capi.Context(60_000, max_frames=2)
dirty_seen = set()
dirty = [0]


def check_clean(it, tag):
    """FUZZ_CHECK_CLEAN=1: what every call must leave behind in slot 0 - occupancy flags zero, the cell table's keys and
    counters zero, its minima all ones - checked after every call (slow: the buffers are downloaded)."""
    if not os.environ.get("FUZZ_CHECK_CLEAN"):
        return
    for name, dt, want in (("cflag", np.uint8, 0), ("hcount", np.uint32, 0), ("hmin", np.uint32, 0xFFFFFFFF), ("hkey", np.uint64, 0),
                           ("cflag@1", np.uint8, 0), ("hcount@1", np.uint32, 0), ("hmin@1", np.uint32, 0xFFFFFFFF), ("hkey@1", np.uint64, 0)):
        if name.endswith("@1") and tag != "batch of two":
            continue
        buf = ctx.debug_buffer(name, dt)
        bad = np.flatnonzero(buf != want)
        if len(bad) and name not in dirty_seen:
            dirty_seen.add(name)
            dirty[0] += 1
            print("DIRTY after case %d %s: %s has %d entries off (first at %d, last at %d)" % (it, tag, name, len(bad), bad[0], bad[-1]), flush=True)
            sys.exit(2)   # (a later frame would read cells without points: stop here)
t0 = time.time()
fails = 0
for it in range(cases):
    kind = kinds[it % len(kinds)]
    n = int(rng.choice([1, 2, 3, 7, 33, 64, 65, 200, 1000, 5000, 20000, 50000]))
    scale = float(rng.choice([0.05, 0.5, 2.0, 20.0]))
    leaf = float(rng.choice([0.01, 0.02, 0.05, 0.2]))
    tol = float(rng.choice([0.02, 0.05, 0.1, 0.4, -0.1]))
    mn = int(rng.choice([1, 2, 5])); mx = int(rng.choice([20, 20000, 10**9]))
    if abs(tol) / scale > 0.2 or kind in ("dups", "lattice", "line"):
        n = min(n, 2000)        # the CPU oracle's radius search is quadratic when everything is within the radius
    pts = cloud(kind, n, scale)
    p = dict(leaf=leaf, tol=tol, min_cluster=mn, max_cluster=mx)
    if it < start:
        dry(it, kind, n, scale, pts)
        continue
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", it, kind, n, scale, p, flush=True)
    ctx.set_params(publish_clouds=1, **p)
    try:
        got = ctx.process_xyz(pts)
        ref = O.process_xyz(pts, **p)
        ok = (got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes() and np.array_equal(got["offsets"], ref["offsets"])
              and np.array_equal(got["indices"], ref["indices"]) and got["passthrough"] == ref["passthrough"])
    except capi.PcopError as e:
        ok = e.status == -7 and scale / abs(tol) > 1e5     # PCOP_ERR_RANGE is the only legitimate refusal here
    check_clean(it, "chain")
    if not ok:
        fails += 1
        print("CHAIN MISMATCH", it, kind, n, scale, p, flush=True)
    if it % 5 == 1 and n >= 3:       # a cloud declared not dense, with NaN / Inf rows: the voxel grid drops them
        q = pts.copy()
        q[::3, rng.integers(0, 3)] = np.nan
        if n > 5:
            q[1::5, 0] = np.inf
        fin = np.isfinite(q).all(axis=1)
        gerr, rerr = None, False
        try:
            got = ctx.process_xyz(q, is_dense=False)
        except capi.PcopError as e:
            gerr = e.status
        try:
            ref = O.process_xyz(q, is_dense=False, **p)
        except ValueError:
            rerr = True       # a passthrough frame (grid overflow) clusters the input itself, NaNs included: refused by both
        if gerr is not None or rerr:
            ok = gerr is not None and (rerr or gerr == -7)
        else:
            ok = (got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes() and np.array_equal(got["offsets"], ref["offsets"])
                  and np.array_equal(got["indices"], ref["indices"]))
        check_clean(it, "nondense chain")
        if not ok:
            fails += 1
            print("NONDENSE MISMATCH", it, kind, n, scale, p, flush=True)
    if it % 7 == 2:                  # the two stage calls, and two frames of different sizes in one batch
        vg, ptg = ctx.voxel_downsample(pts)
        vr, ptr_ = O.voxel_downsample(pts, leaf)
        okv = vg.tobytes() == vr.tobytes() and ptg == ptr_
        check_clean(it, "voxel stage")
        try:
            cg = ctx.euclidean_clusters(pts)
            cr = O.euclidean_clusters(pts, tol, mn, mx)
            okc = np.array_equal(cg[0], cr[0]) and np.array_equal(cg[1], cr[1])
        except capi.PcopError as e:
            okc = e.status == -7
        check_clean(it, "cluster stage")
        other = cloud(kinds[(it + 1) % len(kinds)], max(1, n // 3), scale)
        try:
            ctx.submit_batch_host([pts, other])
            rb = ctx.wait_batch()
            okb = True
            for fr, src in zip(rb, (pts, other)):
                rr = O.process_xyz(src, **p)
                okb = okb and fr["voxel_xyz"].tobytes() == rr["voxel_xyz"].tobytes() and np.array_equal(fr["offsets"], rr["offsets"]) and np.array_equal(fr["indices"], rr["indices"])
        except capi.PcopError as e:
            okb = e.status == -7
        check_clean(it, "batch of two")
        if not (okv and okc and okb):
            fails += 1
            print("STAGE/BATCH MISMATCH", it, kind, n, scale, p, okv, okc, okb, flush=True)
    if it % 3 == 0 and n >= 33 and os.environ.get("PCOP_CELL_PATH") != "hash":      # outlier removal on the same cloud (needs the dense grid's buffers)
        k = int(rng.choice([1, 4, 15, 31]))
        if n >= k + 1:
            if os.environ.get("FUZZ_VERBOSE"):
                print("  sor k", k, flush=True)
            g = ctx.statistical_outlier_removal(pts, k, 1.0)
            check_clean(it, "sor")
            r = O.sor(pts, k, 1.0)
            if not (g["distances"].tobytes() == r["distances"].tobytes() and g["threshold"] == r["threshold"] and g["xyz"].tobytes() == r["xyz"].tobytes()):
                fails += 1
                print("SOR MISMATCH", it, kind, n, scale, k, flush=True)
    if it % 4 == 0:                  # occupancy step with limits cutting through the cloud
        po = O.occ_params(-0.6 * scale, 0.7 * scale, -0.5 * scale, 0.8 * scale, -scale, scale, float(rng.choice([0.013, 0.05, 0.15])) * max(scale, 0.1), 0.5)
        if po.grid_width * po.grid_height <= (1 << 22):
            if os.environ.get("FUZZ_VERBOSE"):
                print("  occ", po.grid_width, po.grid_height, po.block_size, flush=True)
            q = pts.copy()
            if n > 10:
                q[::7, 1] = np.nan; q[::11, 0] = np.nan
            g = ctx.occupancy_filter(q, capi.OccParams(po.x_min, po.x_max, po.y_min, po.y_max, po.z_min, po.z_max, po.block_size, po.dev_percent, po.grid_width, po.grid_height))
            check_clean(it, "occupancy")
            r = O.occupancy(q, po)
            if not (g["xyz"].tobytes() == r["xyz"].tobytes() and np.array_equal(g["counts"], r["counts"]) and np.array_equal(g["grid"], r["grid"])
                    and np.array_equal(g["row_avg"], r["row_avg"]) and g["index_errors"] == r["index_errors"]):
                fails += 1
                print("OCC MISMATCH", it, kind, n, scale, flush=True)
    if it % 6 == 3 and n >= 3 and os.environ.get("PCOP_CELL_PATH") != "hash":   # plane removal, and the node's chain in one call
        thr = float(rng.choice([0.005, 0.02, 0.1])) * max(scale, 0.1)
        g = ctx.segment_planes(pts, capi.plane_params(thr))
        check_clean(it, "planes")
        r = O.segment_planes(pts, thr)
        if not (g["xyz"].tobytes() == r["xyz"].tobytes() and g["n_planes"] == r["n_planes"] and g["coeffs"].tobytes() == r["coeffs"].tobytes()
                and g["found_all"] == r["found_all"]):
            fails += 1
            print("PLANES MISMATCH", it, kind, n, scale, thr, flush=True)
        mk = int(rng.choice([0, 4, 15]))
        use_plane = bool(rng.integers(0, 2))
        po = O.occ_params(-0.6 * scale, 0.7 * scale, -0.5 * scale, 0.8 * scale, -scale, scale, 0.15 * max(scale, 0.1), 0.5)
        pc = capi.OccParams(po.x_min, po.x_max, po.y_min, po.y_max, po.z_min, po.z_max, po.block_size, po.dev_percent, po.grid_width, po.grid_height)
        q = pts.copy()
        if n > 10:
            q[::9] = np.nan            # a depth camera's holes: whole rows
        # the stage-by-stage chain of the restatements
        cloud_r = O.occupancy(q, po)["xyz"]
        vox_r, _ = O.voxel_downsample(cloud_r, leaf)
        refuse = mk and 0 < len(vox_r) <= mk          # the reference reads past its k-NN result there: refused
        try:
            gn = ctx.process_node(q, occ=pc, mean_k=mk, stddev_mul=1.0, plane=capi.plane_params(thr) if use_plane else None, is_dense=False)
            if refuse:
                okn = False
            else:
                cur = vox_r
                if mk and len(cur):
                    cur = O.sor(cur, mk, 1.0)["xyz"]
                if use_plane:
                    cur = O.segment_planes(cur, thr)["xyz"]
                ro, ri = O.euclidean_clusters(cur, tol, mn, mx)
                gv = gn["voxel_xyz"] if gn["voxel_xyz"] is not None else np.zeros((0, 3), dtype=f32)
                okn = gv.tobytes() == cur.tobytes() and np.array_equal(gn["offsets"], ro) and np.array_equal(gn["indices"], ri)
        except capi.PcopError as e:
            okn = (refuse and e.status == -2) or e.status == -7
        check_clean(it, "node chain")
        if not okn:
            fails += 1
            print("NODE CHAIN MISMATCH", it, kind, n, scale, p, mk, use_plane, thr, flush=True)
            if os.environ.get("FUZZ_DUMP"):
                np.savez(os.path.join(os.environ["FUZZ_DUMP"], "fuzz_fail_%d.npz" % it), q=q, leaf=leaf, tol=tol, mn=mn, mx=mx, mk=mk, use_plane=use_plane, thr=thr, scale=scale)
        ctx.set_params(publish_clouds=1, **p)
    if it % 20 == 19:
        print("  ... %d cases, %d mismatches, %.0f s" % (it + 1, fails, time.time() - t0), flush=True)
print("cases %d, mismatches %d, %.1f s" % (cases, fails, time.time() - t0))
if dirty[0]:
    print("buffers left dirty: %s" % sorted(dirty_seen))
ctx.close()
sys.exit(1 if fails or dirty[0] else 0)
