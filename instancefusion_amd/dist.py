"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

What is sharded (DESIGN.md "Multi-GPU"):
* default of `bench.py --gpus N`: N replicas (one stream + one map per rank), no data-path
  collective; only the timing barrier / max-over-ranks go through torch.distributed.
* `bench.py --sharded`: one stream into ONE spatially sharded map; the exchanges of a frame are enqueued by libifx.so itself on a RCCL
  communicator (csrc/ifx_comm.hip), torch.distributed only carries the ncclUniqueId to the ranks.
* `owner_of` is the owner function of the spatially sharded map (8 cm voxel -> Morton code -> mod n_ranks), the
  Python twin of ifx_owner_of_point: `ifx_map_upload` and the append kernel of a handle created with n_ranks > 1 keep the surfels it
  selects (instancefusion_amd/sharded.py: OwnerShardedElasticFusion).
* The tracker stays replicated by default (DESIGN.md section 7 has the numbers); option own_track_rows shards its two reductions over the ranks and
  all-reduces the 2 x 29 exact sums inside the library (csrc/ifx_track.hip k_icp_residual_rows; tests/test_dist_cpu.py exercises the same sum over gloo).
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


def _part1by2_10(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.uint32) & np.uint32(0x3FF)
    v = (v | (v << np.uint32(16))) & np.uint32(0x030000FF)
    v = (v | (v << np.uint32(8))) & np.uint32(0x0300F00F)
    v = (v | (v << np.uint32(4))) & np.uint32(0x030C30C3)
    v = (v | (v << np.uint32(2))) & np.uint32(0x09249249)
    return v


def owner_of(pos: np.ndarray, n_ranks: int) -> np.ndarray:
    """Owner rank of each surfel of the spatially sharded map: a pure function of the position it was created at (so that every
    rank can tell who owns a new surfel without asking): 8 cm voxel (f32 arithmetic) -> 30-bit Morton code -> mod n_ranks.  The same
    function as ifx_owner_of_point (instancefusion_amd/csrc/ifx_dev.h), which the kernels and ifx_map_upload use."""
    if n_ranks <= 1:
        return np.zeros(np.asarray(pos).shape[:-1], np.int32)
    p = np.asarray(pos, np.float32)[..., :3]
    v = np.floor(p / np.float32(VOXEL_M)).astype(np.int64) + 512
    code = _part1by2_10(v[..., 0]) | (_part1by2_10(v[..., 1]) << np.uint32(1)) | (_part1by2_10(v[..., 2]) << np.uint32(2))
    return (code % np.uint32(n_ranks)).astype(np.int32)
