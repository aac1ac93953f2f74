"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

What is sharded (DESIGN.md "Multi-GPU"):
* round 1: `bench.py --gpus N` runs N replicas (one stream + one map per rank), no data-path
  collective; only the timing barrier / max-over-ranks go through torch.distributed.
* the surfel-map owner function below (spatial hash: 8 cm voxels -> Morton code -> mod n_ranks,
  SURVEY.md 8e) and the one-shot all-reduce of the 2 x 29-float normal equations are the two
  primitives of the sharded-map mode; they are exercised on CPU with gloo in tests/test_dist_cpu.py.
"""
from __future__ import annotations

import os

import numpy as np

VOXEL_M = 0.08


def env():
    """(rank, local_rank, world_size) as set by torch.distributed.run."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None):
    """Initialises the process group when WORLD_SIZE > 1; returns (rank, local_rank, world, dist_or_None)."""
    rank, local_rank, world = env()
    if world <= 1:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist


def max_over_ranks(value: float, dist, device="cpu") -> float:
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank: int, world: int, seconds_max: float) -> float:
    """Aggregate throughput: units all ranks processed / slowest rank's time."""
    return world * units_per_rank / seconds_max


def _part1by2(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def owner_of(pos: np.ndarray, n_ranks: int) -> np.ndarray:
    """Owner rank of each surfel: a pure function of its position (so that a surfel's owner can be
    found by any rank): 8 cm voxel -> 63-bit Morton code -> mod n_ranks."""
    v = np.floor(np.asarray(pos, np.float64)[..., :3] / VOXEL_M).astype(np.int64) + (1 << 20)
    code = _part1by2(v[..., 0]) | (_part1by2(v[..., 1]) << np.uint64(1)) | (_part1by2(v[..., 2]) << np.uint64(2))
    return (code % np.uint64(max(n_ranks, 1))).astype(np.int32)


def allreduce_normal_equations(icp29: np.ndarray, rgb29: np.ndarray, dist):
    """The only collective of the tracking stage when image tiles are sharded: 2 x 29 floats summed
    over ranks in one message (latency-bound)."""
    buf = np.concatenate([np.asarray(icp29, np.float64), np.asarray(rgb29, np.float64)])
    if dist is None:
        return buf[:29].copy(), buf[29:].copy()
    import torch

    t = torch.from_numpy(buf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.numpy()
    return out[:29].copy(), out[29:].copy()
