"""The N>1 path on CPU: world_size 2, gloo backend (the GPU run uses the same code over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pointcloud_obstacle_processing_amd import capi, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = capi.Params(np.float32(0.02), np.float32(0.1), 5, 20000, 1) if rank == 0 else None
    got = sharding.broadcast_params(p, src=0)
    frames = sharding.frames_of_rank(10, rank, world)
    total, secs = sharding.merge_counts(len(frames), 1.0 + rank)
    q.put((rank, float(got.leaf), float(got.tol), got.min_cluster, got.max_cluster, got.publish_clouds, frames,
           total, secs))
    dist.barrier()
    dist.destroy_process_group()


def test_params_broadcast_and_frame_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = []
    for rank, leaf, tol, mn, mx, pub, frames, total, secs in res:
        assert (np.float32(leaf), np.float32(tol), mn, mx, pub) == (np.float32(0.02), np.float32(0.1), 5, 20000, 1)
        assert all(sharding.frame_owner(f, world) == rank for f in frames)
        assert total == 10.0 and secs == 2.0
        seen += frames
    assert sorted(seen) == list(range(10))                 # every frame exactly once, no overlap


def test_frame_seeds_are_disjoint():
    a = sharding.frame_seeds(3000, 8, 0)
    b = sharding.frame_seeds(3000, 8, 1)
    assert len(set(a) | set(b)) == 16


def test_params_roundtrip():
    p = capi.Params(np.float32(0.015), np.float32(0.4), 5, 20000, 1)
    q = sharding.params_from_bytes(sharding.params_to_bytes(p))
    assert bytes(p) == bytes(q)


# ---------------------------------------------------------------------------------------------
# bench.py as the driver runs it: `python bench.py --gpus N` starts the N ranks itself
# ---------------------------------------------------------------------------------------------
def _bench(args, env=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=e, capture_output=True, text=True,
                         timeout=timeout)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out.returncode, (json.loads(lines[-1]) if lines else None), out.stderr


def test_bench_launcher_starts_two_ranks():
    """`bench.py --gpus 2` (no torchrun around it) = two ranks, the parameters of rank 0 on both, disjoint frame seeds."""
    rc, d, err = _bench(["--gpus", "2", "--backend", "gloo", "--dry", "--batch", "4", "--contexts", "2"])
    assert rc == 0, err
    assert d["n_gpus"] == 2 and d["dry"] is True
    assert len(d["seeds"]) == 2 and len(d["seeds"][0]) == 8 and set(d["seeds"][0]).isdisjoint(d["seeds"][1])
    assert np.float32(d["leaf"]) == np.float32(0.02) and np.float32(d["tol"]) == np.float32(0.1)


def test_bench_refuses_a_world_size_that_does_not_match():
    """A world size other than --gpus is an error (exit code 2), never a silent downgrade to fewer GPUs."""
    rc, d, err = _bench(["--gpus", "2", "--backend", "gloo", "--dry"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc == 2 and d is None and "does not match" in err


def test_bench_failed_rank_fails_the_launcher():
    rc, d, err = _bench(["--gpus", "2", "--backend", "gloo", "--dry", "--config", "no-such-config"])
    assert rc != 0 and d is None
