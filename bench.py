#!/usr/bin/env python3
"""Benchmark of the hot path (VoxelGrid -> radius graph -> Euclidean clusters) on MI355X.

Metric (BASELINE.json): frames/s at a 1M-point cloud.  Workload: config C3 - 1,000,000 points,
80 % ground plane + 20 % on 20 hovering boxes on a 4.5 m field, leaf 0.02 m, tolerance 0.1 m,
cluster size 5..20000.  Every rank keeps F = contexts x batch distinct seeded clouds
resident in HBM as XYZ SoA planes; one "round" pushes all of them through the whole chain once (the
frames of a batch share their kernel launches, the contexts overlap on separate streams) and one
"step" is `--rounds` rounds, so that the driver's 20 steps time about half a second of GPU work.
Cluster indices are written back to pinned host memory inside the timed region.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), frames sharded by rank, the
32-byte pcop_params struct broadcast from rank 0, no data-path collective ("weak" scaling).  Run as
`python bench.py --gpus N` (this script starts the N ranks itself) or under torchrun (RANK / WORLD_SIZE
in the environment); a world size that does not match --gpus is an error, never a silent downgrade.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate


def kernel_alg_bytes(name, n, m, n_cells):
    """Algorithmic (compulsory) bytes of one launch of a kernel for one frame; DESIGN.md 4.
    n = input points, m = voxels."""
    table = {
        "k_minmax": 12 * n,                     # read the xyz planes
        "k_bin_hist": 12 * n,                   # read the xyz planes (with the keys on the context's box it also takes the bounds: no k_minmax)
        "k_bin_scatter": 12 * n + 12 * n,       # xyz planes in, packed xyz records out
        "k_bucket_reduce": 12 * n + 12 * m,     # records in, centroids out
        "k_radix_hist": 12 * n,
        "k_radix_scatter": 24 * n,
        "k_voxel_reduce": 12 * n + 12 * m,
        "k_cell_insert": 20 * m,
        "k_cell_place": 36 * m,
        "k_cell_mark": 16 * m,
        "k_cell_count": 12 * m,
        "k_cell_starts": 28 * n_cells,
        "k_cell_union": 16 * m, "k_cell_hook": 16 * m, "k_cgrid_union": 16 * m, "k_cgrid_hook": 16 * m,  # reads 12 M, writes 4 M
        "k_cell_flatten": 8 * m,
        "k_csr_hist": 8 * m, "k_csr_scatter": 8 * m,
        "k_kept_apply": 4 * m,
        "k_slot_count": 16 * n_cells, "k_slot_apply": 16 * n_cells, "k_cell_compress": 16 * n_cells,
        "k_emit": 28 * m,
    }
    return table.get(name, 0)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=10, help="rounds (every resident frame once) per step")
    ap.add_argument("--batch", type=int, default=32, help="frames per context, processed in the same launches (32: + 6 %% over 16 in one run on one box, for twice a batch's latency; BASELINE.md 5)")
    ap.add_argument("--contexts", type=int, default=3, help="contexts (streams) in flight per GPU")
    ap.add_argument("--feeders", default="threads", choices=["threads", "loop"], help="one host thread per context (default) or one loop over the contexts")
    ap.add_argument("--rotate", action="store_true", help="hand a context's resident clouds to its slots shifted by one every round (default: a slot sees the same cloud every round); "
                                                      "for measurements of anything that carries state in a slot from batch to batch (DESIGN.md 7.6)")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (and the parity check that uses its results)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 (uniform-random) leg")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry", action="store_true", help="no GPU work: the launcher, the rendezvous, the parameter broadcast and the frame sharding only")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun around it.  The parent touches neither torch nor HIP.
# ---------------------------------------------------------------------------------------------------------
def launch_workers(args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--worker"] + ["--worker"]
        # (rank 0's stdout carries the JSON line; whatever another rank prints on stdout joins every rank's stderr on ours)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode
    line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if rc or not line:
        sys.stderr.write("bench.py: a rank failed (rc %s); rank 0 printed:\n%s\n" % (rc, out0))
        sys.exit(rc or 1)
    print(line[-1])
    sys.exit(0)


def latest_pmc_traffic():
    """The newest profiles/rNN_pmc_traffic.json (the PMC passes of this command are a separate rocprofv3 run)."""
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        mt = re.match(r"r(\d+)_pmc_traffic\.json$", os.path.basename(f))
        if mt and (best is None or int(mt.group(1)) > best[0]):
            best = (int(mt.group(1)), f)
    return best[1] if best else None


def pmc_traffic_of(kernel, batch):
    """HBM bytes per launch of `kernel` from the newest tracked PMC passes.  A kernel may have several template instances in the
    file (the once-per-context order self-check launches `k_bin_scatter<2u, false>`): the one a steady batch launches is the one
    with the most launches (`launches`; older files without the field: the one that moves the most bytes)."""
    f = latest_pmc_traffic()
    if not f:
        return None, None
    try:
        best = None
        for k, v in json.load(open(f)).items():
            if k.split("::")[-1].split("<")[0].split("(")[0] != kernel or v.get("batch") != batch:
                continue
            rank = (v.get("launches", 0), v["hbm_bytes_per_launch"])
            if best is None or rank > best[0]:
                best = (rank, k, v["hbm_bytes_per_launch"])
        if best:
            return best[2], "profiles/%s, instance %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, 2*FETCH+WRITE)" % (os.path.basename(f), best[1])
    except Exception:
        pass
    return None, None


def tracked_launch_us(kernel, batch, solo):
    """Average launch duration of `kernel` in the newest tracked rocprofv3 kernel-trace summary (profiles/rNN_kernel_summary_batchB[_solo].txt)."""
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_summary_batch%d%s.txt" % (batch, "_solo" if solo else ""))):
        mt = re.match(r"r(\d+)_", os.path.basename(f))
        if mt and (best is None or int(mt.group(1)) > best[0]):
            best = (int(mt.group(1)), f)
    if not best:
        return None, None
    try:
        for ln in open(best[1]):
            mt = re.match(r"(?:void )?pcop::(\w+)\S*.*?\s+([0-9.]+)\s+\d+\s+[0-9.]+\s*$", ln)
            if mt and mt.group(1) == kernel:
                return float(mt.group(2)), "profiles/" + os.path.basename(best[1])
    except Exception:
        pass
    return None, None


def same_result(got, ref, np):
    return (got["voxel_xyz"] is not None and got["voxel_xyz"].tobytes() == ref["voxel_xyz"].tobytes()
            and np.array_equal(got["offsets"], ref["offsets"]) and np.array_equal(got["indices"], ref["indices"]))


def worker(args):
    import numpy as np
    import torch  # first: its bundled HIP runtime must be the one libpcop.so binds to
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PCOP_BENCH_SAME_GPU") == "1":   # rehearsal of the N > 1 path on a one-GPU box (with --backend gloo): every rank on device 0
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: WORLD_SIZE (%d) does not match --gpus (%d)\n" % (world, args.gpus))
        sys.exit(2)
    on_gpu = not args.dry
    bound_cpus = 0
    if on_gpu:
        torch.cuda.set_device(local_rank)
        if os.environ.get("PCOP_BENCH_NO_BIND") != "1":
            # this rank's threads and pinned buffers next to its GPU (8 ranks on a two-socket host would otherwise run anywhere)
            from pointcloud_obstacle_processing_amd import capi as _capi
            bound_cpus = _capi.bind_thread_near_device(local_rank)
    # PCOP_BENCH_FORCE_DIST=1 exercises the collective path (init, broadcast, barrier, all-reduce) with one rank too
    use_dist = world > 1 or os.environ.get("PCOP_BENCH_FORCE_DIST") == "1"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    from pointcloud_obstacle_processing_amd import capi, sharding, synth

    # ---- parameters: rank 0 decides, everyone else receives the 32-byte struct ----------------------
    p = synth.params_of(args.config)
    params = capi.Params(np.float32(p["leaf"]), np.float32(p["tol"]), p["min_cluster"], p["max_cluster"], 0)
    if use_dist:
        params = sharding.broadcast_params(params if rank == 0 else None, src=0, device=dev if args.backend == "nccl" else None)

    # ---- frames: F per rank, seeds sharded by rank ----------------------------------------------------
    B, NC = args.batch, args.contexts
    F = B * NC
    n_pts = synth.CONFIGS[args.config]["n"]
    seeds = sharding.frame_seeds(base_seed=3000, frames_per_rank=F, rank=rank)

    def fence():
        if use_dist:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    if args.dry:   # launcher / rendezvous / sharding only
        all_seeds = [None] * world
        if use_dist:
            dist.all_gather_object(all_seeds, seeds)
        else:
            all_seeds = [seeds]
        fence()
        t0 = time.perf_counter()
        fence()
        dt = max(time.perf_counter() - t0, 1e-9)
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "dry run: no GPU work", "value": 0.0, "unit": "frames/s", "n_gpus": world, "dry": True,
                              "leaf": float(params.leaf), "tol": float(params.tol), "seeds": all_seeds, "frames_per_step": F * world * args.rounds}))
        return

    ctxs, ptrs = [], []
    for ci in range(NC):
        c = capi.Context(n_pts, device=local_rank, max_frames=B)
        c.set_params_struct(params)
        mine = [c.upload_soa(synth.make_cloud(args.config, seed=s)) for s in seeds[ci * B:(ci + 1) * B]]
        ctxs.append(c)
        ptrs.append(([q for q, _ in mine], [n for _, n in mine]))

    # One round = every context's batch processed once.  Submission is rolling: a context's previous batch
    # is collected (results on the host) right before its next one is queued, so the other contexts' batches
    # keep the GPU busy meanwhile; drain() collects what is still in flight.  Every frame counted in a timed
    # region is submitted and collected inside it.
    pending = [False] * NC
    last_out = [None]

    def one_round():
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

    # --feeders threads (the default): one host thread per context, each submitting and collecting its own context's batches - the
    # structure of the C++ pool (host/pcop_pool: a feeder thread per GPU).  What a context's thread does between two batches - the clusters
    # of 32 frames put in PCL's order, the next batch's mailboxes, the graph launch: about half a millisecond - then no longer keeps
    # the other contexts' next batches waiting (the library releases no lock: contexts are independent; ctypes drops the GIL for the call).
    # --feeders loop: the single-threaded rolling submission of rounds 1-3.
    def run_rounds(nrounds):
        if args.feeders == "loop" or NC == 1:
            for _ in range(nrounds):
                one_round()
            return drain()
        import threading
        outs = [None] * NC
        errs = []

        def feed(ci):
            try:
                c, (pp, nn) = ctxs[ci], ptrs[ci]
                for r_ in range(nrounds):
                    k_ = r_ % len(pp) if args.rotate else 0
                    c.submit_batch_device(pp[k_:] + pp[:k_], nn[k_:] + nn[:k_])
                    outs[ci] = c.wait_batch(want_arrays=False)[-1]
            except Exception as ex:   # (reported by the main thread)
                errs.append(ex)
        th = [threading.Thread(target=feed, args=(ci,)) for ci in range(NC)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        if errs:
            raise errs[0]
        return outs[-1]

    last = run_rounds(args.warmup * args.rounds)
    fence()
    t0 = time.perf_counter()
    last = run_rounds(args.steps * args.rounds)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = world * args.steps * args.rounds * F
    fps = frames / dt
    m_vox = last["num_voxels"]

    # ---- parity of the configuration that was timed: one more (untimed) round of context 0 with the voxel cloud published, the
    #      same device-resident batch through the same launch sequence (keys on the context's box, streaming voxel path, the tile
    #      size of the timed run), frames 0 and 1 (N > 1: frame 0 of every rank) against the CPU restatement ----
    def timed_path_parity(nframes, refs=None):
        from oracle import oracle as O
        c0 = ctxs[0]
        pub = capi.Params(params.leaf, params.tol, params.min_cluster, params.max_cluster, 1)
        c0.set_params_struct(pub)
        c0.submit_batch_device(*ptrs[0])
        got = c0.wait_batch(want_arrays=True)
        st = c0.debug_state()
        c0.set_params_struct(params)
        bad = []
        for i in range(min(nframes, len(got))):
            ref = refs[i] if refs and i < len(refs) else O.process_xyz(synth.make_cloud(args.config, seed=seeds[i]), **p)
            if not same_result(got[i], ref, np):
                bad.append(seeds[i])
        return bad, {"voxel_path": "stream" if st.vpath else "sort", "keys_on_context_box": bool(st.sbox), "tile_points": int(st.tile),
                     "frames_per_launch": len(got), "buckets_frame0": int(st.nbuckets)}

    parity_all = None
    if use_dist and on_gpu and not args.no_cpu_baseline:
        from oracle import oracle as O
        if rank == 0:
            O.build()
        dist.barrier()
        bad0, _ = timed_path_parity(1)
        t = torch.tensor([0 if bad0 else 1], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        parity_all = bool(int(t.item()))

    result = None
    rc = 0
    if rank == 0:
        # ---- single-frame latency (one context, nothing else on the GPU) -----------------------------
        c0, ptr0, n0 = ctxs[0], ptrs[0][0][0], ptrs[0][1][0]
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
        n_cells = 0
        for _ in range(10):
            prof.submit_batch_device(pp, nn)
            prof.wait_batch(want_arrays=False)
            n_cells = prof.debug_state().n_cells
            seen = {}
            for nm, ms in prof.stage_times():
                occ = seen.get(nm, 0)
                seen[nm] = occ + 1
                a = acc.setdefault((nm, occ), [0.0, 0])
                a[0] += ms
                a[1] += 1
        prof.close()
        by_name = {}
        for (nm, occ), (ms, cnt) in acc.items():
            by_name.setdefault(nm, []).append(ms / cnt)
        totals = {nm: sum(v) for nm, v in by_name.items()}

        def roof(nm):
            ls = by_name[nm]
            ms = sum(ls) / len(ls)   # average duration of one launch (a launch covers B frames)
            nbytes = B * kernel_alg_bytes(nm, n0, m_vox, n_cells)
            return ms, nbytes, (nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0)

        # the dominant kernel: the largest share of the batch; two kernels within 3 % of each other (k_bin_scatter and
        # k_bucket_reduce trade places from run to run) are told apart by their algorithmic bytes - the one that moves more
        # is the one the HBM bound is about
        frame_bytes = 12 * n0 + 28 * m_vox
        kern = [nm for nm in totals if nm.startswith("k_")]
        top = max(totals[nm] for nm in kern)
        dom = max((nm for nm in kern if totals[nm] >= 0.97 * top), key=lambda k: kernel_alg_bytes(k, n0, m_vox, n_cells))
        dom_ms, dom_own_bytes, own_achieved = roof(dom)
        total_ms = sum(totals.values())
        by_kernel = []
        for nm in sorted(totals, key=lambda k: -totals[k]):
            if not nm.startswith("k_") or totals[nm] < 0.05 * total_ms:
                continue
            ms, nbytes, gbs = roof(nm)
            by_kernel.append({"kernel": nm, "launches_per_batch": len(by_name[nm]), "avg_launch_ms": round(ms, 5),
                              "own_alg_bytes_per_launch": nbytes, "own_achieved_GBs": round(gbs, 1), "own_frac": round(gbs / HBM_PEAK_GBS, 4)})
        # HBM bytes per launch of the dominant kernel: from the newest tracked PMC passes of this command (a separate
        # rocprofv3 run: counters cannot be collected inside this process), when they were taken at this batch size
        traffic, traffic_source = pmc_traffic_of(dom, B)
        # the same kernel's launch duration in the tracked rocprofv3 kernel traces: one context alone, and the three contexts of
        # the timed run sharing the GPU
        solo_us, solo_src = tracked_launch_us(dom, B, solo=True)
        shared_us, shared_src = tracked_launch_us(dom, B, solo=False)

        result = {
            "metric": "frames/sec at 1M-pt cloud, 1/2/4/8 MI355X; HBM GB/s vs roofline",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-pt ground-plane + 20 boxes (L=4.5 m), leaf %g m, tol %g m, clusters %d..%d"
                       % (args.config, n0, p["leaf"], p["tol"], p["min_cluster"], p["max_cluster"]),
                       "frames_in_flight_per_gpu": F, "batch_per_launch": B, "contexts_per_gpu": NC, "host_feeders": ("one thread per context" if args.feeders == "threads" and NC > 1 else "one loop"), "rounds_per_step": args.rounds, "clouds_rotate_through_slots": bool(args.rotate), "cpus_bound_near_gpu": bound_cpus,
                       "frames_per_step": F * world * args.rounds, "timed_region_s": dt, "voxels_per_frame": int(m_vox),
                       "input": "XYZ SoA resident in HBM", "parallelism": "frame-sharded, %d rank(s)" % world},
            "latency_ms_single_frame": lat_ms,
            # the frame's algorithmic bytes (12 N + 28 M) x the frames one launch covers, over the dominant kernel's
            # average launch duration.  The duration is measured with HIP events on the profiling context's own stream while
            # that context has the GPU to itself ("solo"); profiles/rNN_kernel_summary_batch16.txt holds rocprofv3's figure for
            # the same kernel with the three contexts of the timed run sharing the GPU.
            "roofline": {"bound": "hbm", "kernel": dom, "launches_per_batch": len(by_name[dom]),
                         "achieved": frame_bytes * B / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": frame_bytes * B / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes_per_frame": frame_bytes, "alg_bytes_per_launch": frame_bytes * B, "frames_per_launch": B,
                         "avg_launch_ms": dom_ms, "duration_source": "HIP events, one context alone on the GPU",
                         "kernel_own_alg_bytes_per_launch": dom_own_bytes, "kernel_own_achieved": own_achieved, "kernel_own_frac": own_achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_over_alg": (traffic / (frame_bytes * B)) if traffic else None,
                         "traffic_over_kernel_own_alg": (traffic / dom_own_bytes) if traffic and dom_own_bytes else None,
                         # the same ratio with the launch duration rocprofv3 recorded (tracked files, reproducible from profiles/)
                         "frac_solo_tracked": (frame_bytes * B / (solo_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if solo_us else None,
                         "avg_launch_us_solo_tracked": solo_us, "solo_tracked_source": solo_src,
                         "frac_shared": (frame_bytes * B / (shared_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if shared_us else None,
                         "avg_launch_us_shared_tracked": shared_us, "shared_tracked_source": shared_src,
                         "peak_measured_copy": 6290.0},
            "roofline_by_kernel": by_kernel,
            "frame_roofline": {"alg_bytes_per_frame": frame_bytes,
                               "achieved_GBs_pipelined": frame_bytes * fps / world / 1e9,
                               "frac_pipelined": frame_bytes * fps / world / 1e9 / HBM_PEAK_GBS,
                               "achieved_GBs_single_frame": frame_bytes / (lat_ms * 1e-3) / 1e9},
            "kernel_ms": {nm: round(t, 5) for nm, t in sorted(totals.items(), key=lambda kv: -kv[1])},
        }

        # ---- end to end: the same pipeline fed from pinned host memory (12 MB per frame over PCIe inside the timed region) ----
        if not args.no_end_to_end and world == 1:   # (N = 1 only, like the CPU leg: the other ranks would only wait)
            import ctypes as C
            host = [torch.from_numpy(np.ascontiguousarray(synth.make_cloud(args.config, seed=s).T.reshape(-1))).pin_memory() for s in seeds[:4]]
            hp = (C.c_void_p * B)(*[C.c_void_p(host[i % len(host)].data_ptr()) for i in range(B)])
            cnt = (C.c_size_t * B)(*[n0] * B)
            res = (capi.Result * B)()

            def h_submit(c):
                c._chk(c.lib.pcop_submit_batch(c.h, hp, capi.LAYOUT_SOA, cnt, B, 1, capi.MEM_HOST))

            def h_wait(c):
                c._chk(c.lib.pcop_wait_batch(c.h, res, B))

            for c in ctxs:
                h_submit(c)
                h_wait(c)
            e_rounds = max(4, 384 // F)
            for c in ctxs:
                h_submit(c)
            t1 = time.perf_counter()
            for _ in range(e_rounds):
                for c in ctxs:
                    h_wait(c)
                    h_submit(c)
            edt = time.perf_counter() - t1
            for c in ctxs:
                h_wait(c)
            efps = e_rounds * F / edt
            result["end_to_end"] = {"value": efps, "unit": "frames/s", "frames": e_rounds * F,
                                    "input": "XYZ SoA in pinned host memory; the H2D copy of 12 MB per frame is inside the timed region",
                                    "pcie_GBs": efps * 12.0 * n0 / 1e9}

        # ---- secondary leg (north star: "uniform-random + planar-ground clouds"): C2, 100 k uniform points, leaf 0.05 / tol 0.2, the same
        #      contexts x batch, device-resident, its own short timed region; never part of `value` ----
        if not args.no_secondary and world == 1:
            p2 = synth.params_of("C2")
            par2 = capi.Params(np.float32(p2["leaf"]), np.float32(p2["tol"]), p2["min_cluster"], p2["max_cluster"], 0)
            n2 = synth.CONFIGS["C2"]["n"]
            seeds2 = [7000 + i for i in range(F)]
            c2, ptr2 = [], []
            for ci in range(NC):
                c = capi.Context(n2, device=local_rank, max_frames=B)
                c.set_params_struct(par2)
                mine = [c.upload_soa(synth.make_cloud("C2", seed=sd)) for sd in seeds2[ci * B:(ci + 1) * B]]
                c2.append(c)
                ptr2.append(([q for q, _ in mine], [n for _, n in mine]))
            pend2 = [False] * NC

            def round2():
                for ci, c in enumerate(c2):
                    if pend2[ci]:
                        c.wait_batch(want_arrays=False)
                    c.submit_batch_device(*ptr2[ci])
                    pend2[ci] = True

            def drain2():
                for ci, c in enumerate(c2):
                    if pend2[ci]:
                        c.wait_batch(want_arrays=False)
                        pend2[ci] = False

            for _ in range(20):
                round2()
            drain2()
            torch.cuda.synchronize()
            r2 = 200
            t1 = time.perf_counter()
            for _ in range(r2):
                round2()
            drain2()
            torch.cuda.synchronize()
            sdt = time.perf_counter() - t1
            sec = {"workload": "C2: %d-pt uniform-random cloud in the field box, leaf %g m, tol %g m, clusters %d..%d" % (n2, p2["leaf"], p2["tol"], p2["min_cluster"], p2["max_cluster"]),
                   "value": r2 * F / sdt, "unit": "frames/s", "frames": r2 * F, "timed_region_s": sdt, "contexts_per_gpu": NC, "batch_per_launch": B,
                   "input": "XYZ SoA resident in HBM"}
            if not args.no_cpu_baseline:   # parity of this leg's timed configuration: frame 0 of context 0's batch
                from oracle import oracle as O
                O.build()
                c2[0].set_params_struct(capi.Params(par2.leaf, par2.tol, par2.min_cluster, par2.max_cluster, 1))
                c2[0].submit_batch_device(*ptr2[0])
                got2 = c2[0].wait_batch(want_arrays=True)
                ref2 = O.process_xyz(synth.make_cloud("C2", seed=seeds2[0]), **p2)
                ok2 = same_result(got2[0], ref2, np)
                sec["parity"] = "bit-exact vs the CPU restatement on frame 0 of the timed batch" if ok2 else "MISMATCH"
                sec["voxels_per_frame"] = int(got2[0]["num_voxels"])
                sec["clusters_kept"] = int(len(got2[0]["offsets"]) - 1)
                sec["alg_bytes_per_frame"] = 12 * n2 + 28 * sec["voxels_per_frame"]
                sec["frac_pipelined"] = sec["alg_bytes_per_frame"] * sec["value"] / 1e9 / HBM_PEAK_GBS
                if not ok2:
                    rc = 1
            for c in c2:
                c.close()
            result["secondary"] = sec

        # ---- CPU baseline: the oracle (a restatement of the PCL path, not PCL) on this box's host cores; its results for the
        #      first frames are then compared with what the GPU gives for the same clouds (outside every timed region) ----
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as O
            O.build()
            clouds = [synth.make_cloud(args.config, seed=s) for s in seeds[: min(2, len(seeds))]]
            refs = []
            t1 = time.perf_counter()
            done = 0
            while done < args.cpu_frames and time.perf_counter() - t1 < 25.0:
                r = O.process_xyz(clouds[done % len(clouds)], **p)
                if done < len(clouds):
                    refs.append(r)
                done += 1
            cdt = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": done / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                                      "sample": "%d frames of %s (%d pts) through oracle/pcl_restatement.cpp "
                                                "(kd-tree built twice, sorted radius search, serial BFS), 1 thread, "
                                                "host has %d cores" % (done, args.config, n0, os.cpu_count())}
            # (a) the timed configuration itself (batched, device-resident, keys on the context's box)
            bad, how = timed_path_parity(2, refs)
            # (b) the single-frame entry point on a fresh context (the sort voxel path, with its bounds pass)
            chk = capi.Context(n_pts, device=local_rank)
            chk.set_params(leaf=p["leaf"], tol=p["tol"], min_cluster=p["min_cluster"], max_cluster=p["max_cluster"], publish_clouds=1)
            bad1 = [seeds[i] for i, ref in enumerate(refs) if not same_result(chk.process_xyz(clouds[i]), ref, np)]
            chk.close()
            # SURVEY 8(d), secondary work figure of the neighbour pass: 2 E directed pair tests a frame (E = pairs within the
            # tolerance among the frame's voxel centroids; counted here with SciPy's cKDTree on the oracle's cloud, outside every timed region)
            try:
                from scipy.spatial import cKDTree
                vox = refs[0]["voxel_xyz"].astype(np.float64)
                tr = cKDTree(vox)
                e_pairs = (int(tr.count_neighbors(tr, float(p["tol"]))) - len(vox)) // 2
                result["pair_tests"] = {"pairs_per_frame": e_pairs, "directed_tests_per_frame": 2 * e_pairs,
                                        "pair_tests_per_s": 2.0 * e_pairs * fps / world, "counted_on": "seed %d, cKDTree.count_neighbors (d <= tol in float64)" % seeds[0]}
            except Exception as ex:   # (SciPy missing: the figure is left out, nothing else depends on it)
                result["pair_tests"] = {"error": str(ex)}
            result["parity_checked"] = 2
            result["parity"] = ("bit-exact vs the CPU restatement (centroids, cluster offsets, indices) on frames 0 and 1 of the TIMED configuration: "
                                "context 0's batch of %d device-resident frames, %s voxel path, keys on the context's box: %s, %d points per tile"
                                % (how["frames_per_launch"], how["voxel_path"], how["keys_on_context_box"], how["tile_points"])) if not bad else "MISMATCH on seeds %s (timed configuration)" % bad
            result["parity_timed_path"] = how
            result["parity_single_frame_checked"] = len(refs)
            result["parity_single_frame"] = "bit-exact vs the CPU restatement (fresh single-frame context, sort voxel path)" if not bad1 else "MISMATCH on seeds %s" % bad1
            if bad or bad1:
                rc = 1

    if rank == 0 and parity_all is not None:
        result["parity_checked"] = world
        result["parity"] = ("bit-exact vs the CPU restatement on frame 0 of every rank's TIMED configuration (batched, device-resident, keys on the context's box)" if parity_all
                            else "MISMATCH on some rank's first frame")
        if not parity_all:
            rc = 1
    for c in ctxs:
        c.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))
    if rc:
        sys.stderr.write("bench.py: GPU results differ from the CPU restatement\n")
        sys.exit(rc)


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        sys.exit(2)
    if args.gpus > 1 and "RANK" not in os.environ and not args.worker:
        launch_workers(args)   # does not return
    worker(args)


if __name__ == "__main__":
    main()
