#!/usr/bin/env python3
"""Benchmark of the hot path (VoxelGrid -> radius graph -> Euclidean clusters) on MI355X.

Metric (BASELINE.json): frames/s at a 1M-point cloud.  Workload: config C3 — 1,000,000 points,
80 % ground plane + 20 % on 20 hovering boxes on a 4.5 m field, leaf 0.02 m, tolerance 0.1 m,
cluster size 5..20000.  One "step" = one batch of F frames (distinct seeded
clouds, already resident in HBM as XYZ SoA planes) pushed through the whole chain, F contexts in
flight per GPU (contexts x batch: the frames of a batch share their kernel launches); cluster
indices are written back to pinned host memory inside the timed region.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), frames sharded by rank,
the 32-byte pcop_params struct broadcast from rank 0, no data-path collective ("weak" scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate


def kernel_alg_bytes(name, occurrence, n, m, n_cells):
    """Algorithmic (compulsory) bytes of one launch of a kernel of the frame; DESIGN.md §Roofline.
    n = input points, m = voxels.  `occurrence` = index of this launch among launches of the same name."""
    if name == "k_minmax":
        return 12 * n
    if name == "k_bin_hist":
        return 12 * n                   # read the xyz planes
    if name == "k_bin_scatter":
        return 12 * n + 12 * n          # xyz planes in, packed xyz records out
    if name == "k_bucket_reduce":
        return 12 * n + 12 * m          # records in, centroids out
    if name == "k_radix_hist":
        return 12 * n                   # xyz planes (pass 0) or packed 12-byte records
    if name == "k_radix_scatter":
        return (12 + 12) * n            # read every record once, write it once
    if name == "k_voxel_reduce":
        return 12 * n + 12 * m
    if name == "k_cell_insert":
        return 12 * m + 8 * m
    if name == "k_cell_place":
        return 12 * m + 8 * m + 16 * m
    if name == "k_cell_mark":
        return 12 * m + 4 * m
    if name == "k_cell_count":
        return 4 * m + 8 * m
    if name == "k_cell_starts":
        return 28 * n_cells
    if name in ("k_cell_union", "k_cell_hook", "k_cgrid_union", "k_cgrid_hook"):
        return 12 * m + 4 * m          # neighbour + cluster pass reads 12·M, writes 4·M
    if name == "k_cell_flatten":
        return 8 * m
    if name in ("k_csr_hist", "k_csr_scatter"):
        return 8 * m
    if name in ("k_kept_count", "k_kept_apply"):
        return 4 * m
    if name in ("k_slot_count", "k_slot_apply", "k_cell_compress"):
        return 16 * n_cells
    if name == "k_emit":
        return 28 * m
    return 0


kernel_alg_bytes.voxel_passes = 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="frames per context, processed in the same launches")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) in flight per GPU")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=8)
    args = ap.parse_args()

    import torch  # first: its bundled HIP runtime must be the one libpcop.so binds to
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus), file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local_rank)
    # PCOP_BENCH_FORCE_DIST=1 exercises the RCCL path (init, broadcast, barrier, all-reduce) with one rank too
    use_dist = world > 1 or os.environ.get("PCOP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from pointcloud_obstacle_processing_amd import capi, sharding, synth

    # ---- parameters: rank 0 decides, everyone else receives the 32-byte struct over RCCL ----------
    p = synth.params_of(args.config)
    params = capi.Params(np.float32(p["leaf"]), np.float32(p["tol"]), p["min_cluster"], p["max_cluster"], 0)
    if use_dist:
        params = sharding.broadcast_params(params if rank == 0 else None, src=0, device=torch.device("cuda", local_rank))

    # ---- frames: F per rank, seeds sharded by rank --------------------------------------------------
    B, NC = args.batch, args.contexts
    F = B * NC
    n_pts = synth.CONFIGS[args.config]["n"]
    seeds = sharding.frame_seeds(base_seed=3000, frames_per_rank=F, rank=rank)
    ctxs, ptrs = [], []
    for ci in range(NC):
        c = capi.Context(n_pts, device=local_rank, max_frames=B)
        c.set_params_struct(params)
        mine = [c.upload_soa(synth.make_cloud(args.config, seed=s)) for s in seeds[ci * B:(ci + 1) * B]]
        ctxs.append(c)
        ptrs.append(([p for p, _ in mine], [n for _, n in mine]))

    # One step = every context's batch processed once.  Submission is rolling: a context's previous batch
    # is collected (results on the host) right before its next one is queued, so the other contexts' batches
    # keep the GPU busy meanwhile; drain() collects what is still in flight.  Every frame counted in a timed
    # region is submitted and collected inside it.
    pending = [False] * NC
    last_out = [None]

    def step():
        for ci, (c, (pp, nn)) in enumerate(zip(ctxs, ptrs)):
            if pending[ci]:
                last_out[0] = c.wait_batch(want_arrays=False)[-1]
            c.submit_batch_device(pp, nn)
            pending[ci] = True

    def drain():
        for ci, c in enumerate(ctxs):
            if pending[ci]:
                last_out[0] = c.wait_batch(want_arrays=False)[-1]
                pending[ci] = False
        return last_out[0]

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    last = drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    last = drain()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = world * args.steps * F
    fps = frames / dt
    m_vox = last["num_voxels"]

    result = None
    if rank == 0:
        # ---- single-frame latency (one context, nothing else on the GPU) -----------------------------
        c0, ptr0, n0 = ctxs[0], ptrs[0][0][0], ptrs[0][1][0]
        m_vox = last["num_voxels"]
        lat = []
        for _ in range(20):
            t1 = time.perf_counter()
            c0.submit_device(ptr0, n0)
            c0.wait(want_arrays=False)
            lat.append(time.perf_counter() - t1)
        lat_ms = 1e3 * float(np.median(lat))

        # ---- per-kernel times: HIP events on the context's own stream, same batch size as the timed run ----
        prof = capi.Context(n_pts, device=local_rank, profiling=True, max_frames=B)
        prof.set_params_struct(params)
        pp, nn = ptrs[0]
        for _ in range(3):
            prof.submit_batch_device(pp, nn)
            prof.wait_batch(want_arrays=False)
        acc = {}
        reps = 10
        n_cells = 0
        for _ in range(reps):
            prof.submit_batch_device(pp, nn)
            prof.wait_batch(want_arrays=False)
            st = prof.debug_state()
            n_cells = st.n_cells
            seen = {}
            for nm, ms in prof.stage_times():
                occ = seen.get(nm, 0)
                seen[nm] = occ + 1
                a = acc.setdefault((nm, occ), [0.0, 0])
                a[0] += ms
                a[1] += 1
        per_launch = {k: v[0] / v[1] for k, v in acc.items()}
        by_name = {}
        for (nm, occ), ms in per_launch.items():
            by_name.setdefault(nm, []).append((occ, ms))
        totals = {nm: sum(ms for _, ms in v) for nm, v in by_name.items()}
        dom = max((nm for nm in totals if nm.startswith("k_")), key=lambda k: totals[k])
        dom_launches = by_name[dom]
        dom_ms = sum(ms for _, ms in dom_launches) / len(dom_launches)
        # one launch covers B frames
        dom_bytes = B * sum(kernel_alg_bytes(dom, occ, n0, m_vox, n_cells) for occ, _ in dom_launches) / len(dom_launches)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        frame_bytes = 12 * n0 + 28 * m_vox
        prof.close()
        # the same figures for every kernel that takes at least 5 % of the batch, largest first
        total_ms = sum(totals.values())
        by_kernel = []
        for nm in sorted(totals, key=lambda k: -totals[k]):
            if not nm.startswith("k_") or totals[nm] < 0.05 * total_ms:
                continue
            ls = by_name[nm]
            ms = sum(t for _, t in ls) / len(ls)
            nbytes = B * sum(kernel_alg_bytes(nm, occ, n0, m_vox, n_cells) for occ, _ in ls) / len(ls)
            by_kernel.append({"kernel": nm, "launches_per_batch": len(ls), "avg_launch_ms": round(ms, 5),
                              "alg_bytes_per_launch": nbytes, "achieved_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                              "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        # HBM bytes per launch from the committed PMC passes of this command (profiles/), if they match
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            for k, v in pmc.items():
                if k.split("::")[-1].split("<")[0] == dom and v.get("batch") == B:
                    traffic = v["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

        result = {
            "metric": "frames/sec at 1M-pt cloud, 1/2/4/8 MI355X; HBM GB/s vs roofline",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-pt ground-plane + 20 boxes (L=4.5 m), leaf %g m, tol %g m, clusters %d..%d"
                       % (args.config, n0, p["leaf"], p["tol"], p["min_cluster"], p["max_cluster"]),
                       "frames_in_flight_per_gpu": F, "batch_per_launch": B, "contexts_per_gpu": NC,
                       "frames_per_step": F * world, "voxels_per_frame": int(m_vox),
                       "input": "XYZ SoA resident in HBM", "parallelism": "frame-sharded, %d rank(s)" % world},
            "latency_ms_single_frame": lat_ms,
            "roofline": {"bound": "hbm", "kernel": dom, "launches_per_frame": len(dom_launches),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "avg_launch_ms": dom_ms, "alg_bytes_per_launch": dom_bytes,
                         "frames_per_launch": B, "peak_measured_copy": 6290.0},
            "roofline_by_kernel": by_kernel,
            "frame_roofline": {"alg_bytes_per_frame": frame_bytes,
                               "achieved_GBs_pipelined": frame_bytes * fps / world / 1e9,
                               "frac_pipelined": frame_bytes * fps / world / 1e9 / HBM_PEAK_GBS,
                               "achieved_GBs_single_frame": frame_bytes / (lat_ms * 1e-3) / 1e9},
            "kernel_ms": {nm: round(t, 5) for nm, t in sorted(totals.items(), key=lambda kv: -kv[1])},
        }

        # ---- CPU baseline: the oracle (a restatement of the PCL path, not PCL) on this box's host cores ----
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            O.build()
            clouds = [synth.make_cloud(args.config, seed=s) for s in seeds[: min(2, len(seeds))]]
            t1 = time.perf_counter()
            done = 0
            while done < args.cpu_frames and time.perf_counter() - t1 < 25.0:
                O.process_xyz(clouds[done % len(clouds)], **p)
                done += 1
            cdt = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": done / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                                      "sample": "%d frames of %s (%d pts) through oracle/pcl_restatement.cpp "
                                                "(kd-tree built twice, sorted radius search, serial BFS), 1 thread, "
                                                "host has %d cores" % (done, args.config, n0, os.cpu_count())}

    for c in ctxs:
        c.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
