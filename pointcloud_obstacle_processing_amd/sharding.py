"""Frame-level sharding across GPUs: frames are independent, so rank r of G
processes frames f with f mod G == r.  The only bytes that cross GPUs are the 32-byte
pcop_params struct, broadcast from rank 0 (RCCL when the tensor is on a GPU, gloo on CPU).
There is no data-path collective.
"""
import ctypes as C

import numpy as np

from .capi import Params


def frame_owner(frame_index, world_size):
    return frame_index % world_size


def frames_of_rank(num_frames, rank, world_size):
    """Global frame indices handled by `rank`."""
    return list(range(rank, num_frames, world_size))


def frame_seeds(base_seed, frames_per_rank, rank):
    """Seeds of the synthetic frames a rank keeps in flight (distinct across ranks)."""
    return [base_seed + rank * frames_per_rank + i for i in range(frames_per_rank)]


def params_to_bytes(p):
    assert C.sizeof(Params) == 32
    return np.frombuffer(bytes(p), dtype=np.uint8).copy()


def params_from_bytes(b):
    raw = np.ascontiguousarray(b, dtype=np.uint8).tobytes()
    assert len(raw) == 32
    return Params.from_buffer_copy(raw)


def broadcast_params(params, src=0, device=None):
    """torch.distributed broadcast of the pcop_params POD.  `params` is needed on `src` only."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    if dist.get_rank() == src:
        t = torch.from_numpy(params_to_bytes(params)).to(dev)
    else:
        t = torch.zeros(32, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return params_from_bytes(t.cpu().numpy())


def merge_counts(local_frames, local_seconds):
    """All-reduce of (frames, max seconds) -> whole-job frames/s; used by the benchmark bookkeeping."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    f = torch.tensor([float(local_frames)], dtype=torch.float64, device=dev)
    s = torch.tensor([float(local_seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.MAX)
    return float(f.item()), float(s.item())
