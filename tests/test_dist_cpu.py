"""World-size-2 gloo tests (CPU) of the N > 1 path: timing reduction, aggregate rate, the ICP
normal-equation all-reduce and the spatial-hash owner function."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from instancefusion_amd import dist as ifd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w, d = ifd.init("gloo")
    assert (r, w) == (rank, world) and d is not None
    d.barrier()
    dt = ifd.max_over_ranks(1.0 + rank, d)              # slowest rank decides
    rate = ifd.whole_job_rate(100, w, dt)
    # tile-sharded tracking: each rank owns half of the image rows; the sums must equal the full sums
    rng = np.random.RandomState(7)
    rows = rng.standard_normal((480, 29))
    part = rows[rank::world].sum(0)
    icp, rgb = ifd.allreduce_normal_equations(part, 2 * part, d)
    ok = np.allclose(icp, rows.sum(0)) and np.allclose(rgb, 2 * rows.sum(0))
    q.put((rank, dt, rate, bool(ok)))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, dt, rate, ok in res:
        assert dt == 2.0 and rate == 100.0 and ok


def test_owner_function_partitions_space():
    rng = np.random.RandomState(0)
    pos = rng.uniform(-3, 3, (200000, 3))
    for g in (1, 2, 4, 8):
        own = ifd.owner_of(pos, g)
        assert own.min() >= 0 and own.max() < g
        cnt = np.bincount(own, minlength=g)
        assert cnt.min() > 0.8 * len(pos) / g          # balanced
    # pure function of the voxel: points in one 8 cm voxel share an owner
    base = np.floor(pos[:1000] / ifd.VOXEL_M) * ifd.VOXEL_M
    a = ifd.owner_of(base + 0.01, 8)
    b = ifd.owner_of(base + 0.07, 8)
    assert np.array_equal(a, b)
    assert ifd.env() == (0, 0, 1)


def _minworker(rank, world, port, q):
    import torch

    from instancefusion_amd import sharded

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w, d = ifd.init("gloo")
    rng = np.random.RandomState(100 + rank)
    # 64-bit keys as the rasteriser makes them: depth bits << 32 | slot id; 0xFFFF... = empty pixel
    z = rng.uniform(0.3, 9.0, 5000).astype(np.float32).view(np.uint32).astype(np.uint64)
    keys = (z << np.uint64(32)) | rng.randint(0, 1 << 22, 5000).astype(np.uint64)
    keys[rng.rand(5000) < 0.4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    t = torch.from_numpy(keys.view(np.int64).copy())
    sharded.KeyExchange.reduce_min([t], d)
    q.put((rank, keys, t.numpy().view(np.uint64).copy()))
    d.barrier()
    d.destroy_process_group()


def test_key_image_min_allreduce_gloo():
    """The exchange of the sharded projection: element-wise UNSIGNED 64-bit minimum across ranks (signed all-reduce after
    flipping the sign bit), here over gloo with the key layout of the rasteriser incl. the all-ones 'empty' key."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_minworker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.minimum(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], want) and np.array_equal(res[1][2], want)
    assert (want == np.uint64(0xFFFFFFFFFFFFFFFF)).sum() > 0
