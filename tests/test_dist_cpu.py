"""World-size-2 gloo tests (CPU) of the N > 1 path: timing reduction, aggregate rate, the ICP
normal-equation all-reduce and the spatial-hash owner function."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from instancefusion_amd import dist as ifd


def _allreduce_normal_equations(icp29, rgb29, dist):
    """The one collective a row-tiled tracker would need: 2 x 29 sums in one message.  The sums are exact integers in f64 (DESIGN.md
    "Arithmetic contract"), so an all-reduce(SUM) of them is exact and order-independent.  (The product keeps the tracker replicated.)"""
    import torch

    buf = np.concatenate([np.asarray(icp29, np.float64), np.asarray(rgb29, np.float64)])
    t = torch.from_numpy(buf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.numpy()
    return out[:29].copy(), out[29:].copy()


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
    icp, rgb = _allreduce_normal_equations(part, 2 * part, d)
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


def test_owner_function_equals_the_library():
    """dist.owner_of is the Python twin of ifx_owner_of_point, the function the kernels and ifx_map_upload use (host entry: no GPU needed)."""
    import ctypes as C

    import instancefusion_amd as ifx

    L = ifx.lib()
    rng = np.random.RandomState(3)
    pos = np.concatenate([rng.uniform(-8, 8, (50000, 3)), rng.uniform(-0.2, 0.2, (5000, 3)), np.floor(rng.uniform(-50, 50, (5000, 3))) * 0.08]).astype(np.float32)
    for g in (1, 2, 3, 4, 8):
        out = np.zeros(len(pos), np.int32)
        assert L.ifx_owner_of(pos.ctypes.data_as(C.c_void_p), len(pos), g, out.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(out, ifd.owner_of(pos, g)), g


def _sumworker(rank, world, port, q):
    import torch

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w, d = ifd.init("gloo")
    rng = np.random.RandomState(5)
    # the attribute exchange of the sharded map: every pixel is written by ONE rank (the owner of its winner), zeros elsewhere; the
    # all-reduce is a SUM over int32 words, i.e. a bitwise merge -- also for -0.0, NaN payloads and denormals
    full = rng.standard_normal((4000, 4)).astype(np.float32)
    full[::97] = np.float32(-0.0); full[5::131] = np.nan; full[7::173] = np.float32(1e-42)
    owner = rng.randint(0, world, 4000)
    mine = np.where((owner == rank)[:, None], full, np.float32(0)).astype(np.float32)
    t = torch.from_numpy(mine.view(np.int32).copy())
    d.all_reduce(t, op=d.ReduceOp.SUM)
    # the tracker's collective: exact (integer-valued) f64 sums -> the all-reduce is exact whatever the association order
    part = np.floor(rng.standard_normal((480, 29)) * 2.0 ** 20)[rank::world].sum(0)
    tot = np.floor(np.random.RandomState(5).standard_normal((4000, 4)) * 0 + 0)  # (keeps the stream of `rng` aligned across ranks)
    icp, _ = _allreduce_normal_equations(part, part, d)
    q.put((rank, full.view(np.int32).copy(), t.numpy().copy(), icp))
    d.barrier()
    d.destroy_process_group()


def test_attribute_sum_exchange_and_exact_normal_equations_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sumworker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert np.array_equal(r[2], res[0][1])                       # bit patterns of the merged image == the unsharded image, on both ranks
    rng = np.random.RandomState(5)
    rng.standard_normal((4000, 4)); rng.randint(0, 2, 4000)
    rows = np.floor(rng.standard_normal((480, 29)) * 2.0 ** 20)
    assert np.array_equal(res[0][3], rows.sum(0)) and np.array_equal(res[1][3], rows.sum(0))   # exact, not allclose


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
    # one word behind the image, in the same collective (exchanges 0 / 2 / 4 of the owner-sharded frame): this rank's lowest live creation number -- the reference's
    # "surfel 0" is the lowest of any rank; a rank without a live surfel publishes all ones, the identity of the unsigned minimum
    keys = np.concatenate([keys, np.array([0xFFFFFFFFFFFFFFFF if rank == 1 else 4711], np.uint64)])
    t = torch.from_numpy(keys.view(np.int64).copy())
    sharded.KeyExchange.reduce_min([t], d)
    once = t.numpy().view(np.uint64).copy()
    sharded.KeyExchange.reduce_min([t], d)          # exchange 2 reduces the reduced word again: a minimum of equal values
    assert np.array_equal(t.numpy().view(np.uint64), once)
    q.put((rank, keys, once))
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
    assert want[-1] == 4711 and np.int32(np.uint32(want[-1] & np.uint64(0xFFFFFFFF))) == 4711   # (the kernels read the low word as an int: FIRST_LIVE)


def _rsworker(rank, world, port, q):
    import torch

    from instancefusion_amd import sharded

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w, d = ifd.init("gloo")
    rng = np.random.RandomState(200 + rank)
    W, H = 50, 37                                       # (a lattice of 5 x 4 points; 1850 pixels + the word: 1851 keys, an odd number -> tiles of 926 with one key of slack)
    P = W * H
    z = rng.uniform(0.3, 9.0, P).astype(np.float32).view(np.uint32).astype(np.uint64)
    keys = (z << np.uint64(32)) | rng.randint(0, 1 << 22, P).astype(np.uint64)
    keys[rng.rand(P) < 0.4] = np.uint64(0xFFFFFFFFFFFFFFFF)
    keys = np.concatenate([keys, np.array([0xFFFFFFFFFFFFFFFF if rank == 1 else 4711], np.uint64)])
    n = P + 1
    # ---- op 6 as ifx_comm.hip runs it (comm_keys_rs): MIN by reduce-scatter over `world` tiles of equal size (gloo has no reduce_scatter: one reduce per tile, to the
    # tile's rank), the low words of this rank's tile, an all-gather of them, keys rebuilt from the low words
    tile = (n + world - 1) // world
    padded = np.concatenate([keys, rng.randint(0, 1 << 62, tile * world - n).astype(np.uint64)])   # (what lies in the slack is anything: it never comes back)
    flipped = torch.from_numpy((padded ^ np.uint64(1 << 63)).view(np.int64).copy())
    mine = None
    for g in range(world):
        t = flipped[g * tile:(g + 1) * tile].clone()
        d.reduce(t, dst=g, op=d.ReduceOp.MIN)
        if g == rank:
            mine = t
    low = (mine.numpy().view(np.uint64) ^ np.uint64(1 << 63)).astype(np.uint32)    # k_keys_low
    gathered = [torch.empty(tile, dtype=torch.int32) for _ in range(world)]
    d.all_gather(gathered, torch.from_numpy(low.view(np.int32).copy()))
    allv = np.concatenate([g_.numpy().view(np.uint32) for g_ in gathered])[:n]
    rebuilt = np.where(allv == np.uint32(0xFFFFFFFF), np.uint64(0xFFFFFFFFFFFFFFFF), allv.astype(np.uint64))   # k_keys_from_low
    # ---- the same exchange as a caller-driven transport may run it: the whole MIN, then the depth stripped (sharded._strip_depth)
    t = torch.from_numpy(keys.view(np.int64).copy())
    sharded.KeyExchange.reduce_min([t], d)
    whole_min = t.numpy().view(np.uint64).copy()
    sharded._strip_depth(t)
    stripped = t.numpy().view(np.uint64).copy()
    # ---- option own_lazy_ids: [splat keys | id keys of the lattice | word] in ONE minimum == the lattice of the minimum of the whole id image
    ids = keys[:P].copy(); rng.shuffle(ids)
    lat = np.array([(y * 10) * W + x * 10 for y in range((H + 9) // 10) for x in range((W + 9) // 10)])
    packed = np.concatenate([keys[:P], ids[lat], keys[P:]])
    tp = torch.from_numpy(packed.view(np.int64).copy())
    sharded.KeyExchange.reduce_min([tp], d)
    ti = torch.from_numpy(ids.view(np.int64).copy())
    sharded.KeyExchange.reduce_min([ti], d)
    q.put((rank, whole_min, rebuilt, stripped, tp.numpy().view(np.uint64).copy(), ti.numpy().view(np.uint64)[lat].copy(), len(lat)))
    d.barrier()
    d.destroy_process_group()


def test_index_keys_reduce_scatter_and_lattice_pack_gloo():
    """The two exchange options of the sharded map over gloo with two ranks.  own_key_rs (ifx_owner_exchange op 6): the index keys' MIN as a
    reduce-scatter over equal tiles + an all-gather of the low words gives every rank (uint64) creation number of the winner, empty keys and an all-ones word whole --
    the same as the whole MIN with the depth stripped.  own_lazy_ids: the id keys of the 10 x 10 lattice packed between the splat keys and the word."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rsworker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, whole_min, rebuilt, stripped, packed, lat_of_min, nl in res:
        empty = whole_min == np.uint64(0xFFFFFFFFFFFFFFFF)
        want = np.where(empty, whole_min, whole_min & np.uint64(0xFFFFFFFF))
        assert np.array_equal(rebuilt, want) and np.array_equal(stripped, want), rank
        assert empty.sum() > 100 and want[-1] == 4711
        P = len(whole_min) - 1
        assert nl == 20 and np.array_equal(packed[:P], whole_min[:P]) and np.array_equal(packed[P:P + nl], lat_of_min) and packed[-1] == 4711, rank
    assert np.array_equal(res[0][2], res[1][2])


def _segworker(rank, world, port, q):
    import torch

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w, d = ifd.init("gloo")
    rng = np.random.RandomState(7)                       # the same "unsharded" data on every rank ...
    W, H, NB = 320, 240, 40
    px = rng.randint(0, W * H, 6000)
    box_of = rng.randint(0, NB, 6000)
    owner = (px * 7 + px // 13) % world                   # ... of which this rank sees the pixels whose surfel it owns (one owner per pixel)
    depth = rng.randint(1, 60000, W * H).astype(np.uint16)
    stat_max, stat_sum = rng.randint(0, 200, (6000,)), rng.randint(0, 50, (6000,))

    def boxes(sel):
        b = np.tile(np.array([W, 0, H, 0], np.int32), (NB, 1))       # {minX, maxX, minY, maxY} as k_init_bbox leaves them
        for p, k in zip(px[sel], box_of[sel]):
            x, y = p % W, p // W
            b[k] = (min(b[k, 0], x), max(b[k, 1], x), min(b[k, 2], y), max(b[k, 3], y))
        return b

    mine = owner == rank
    # exchange point 1: maxima negated (k_bbox_flip), ONE int32 MIN, negated back
    b = boxes(mine)
    b[:, 1::2] *= -1
    t = torch.from_numpy(b.reshape(-1).copy())
    d.all_reduce(t, op=d.ReduceOp.MIN)
    merged = t.numpy().reshape(NB, 4).copy()
    merged[:, 1::2] *= -1
    # exchange point 2: the uint16 model depth, disjoint supports, summed as int32 words
    part = np.zeros(W * H, np.uint16)
    sel = np.zeros(W * H, bool); sel[px[mine]] = True
    part[sel] = depth[sel]
    t2 = torch.from_numpy(part.view(np.int32).copy())
    d.all_reduce(t2, op=d.ReduceOp.SUM)
    # exchange point 3: per-instance maximum (MAX) and sum (SUM) of the vote counters
    mx, sm = np.zeros(NB, np.int32), np.zeros(NB, np.int32)
    np.maximum.at(mx, box_of[mine], stat_max[mine].astype(np.int32))
    np.add.at(sm, box_of[mine], stat_sum[mine].astype(np.int32))
    t3, t4 = torch.from_numpy(mx), torch.from_numpy(sm)
    d.all_reduce(t3, op=d.ReduceOp.MAX)
    d.all_reduce(t4, op=d.ReduceOp.SUM)
    seen = np.zeros(W * H, bool); seen[px] = True
    want_depth = np.where(seen, depth, 0).astype(np.uint16)
    wmx, wsm = np.zeros(NB, np.int32), np.zeros(NB, np.int32)
    np.maximum.at(wmx, box_of, stat_max.astype(np.int32))
    np.add.at(wsm, box_of, stat_sum.astype(np.int32))
    ok = (np.array_equal(merged, boxes(np.ones(6000, bool))) and np.array_equal(t2.numpy().view(np.uint16), want_depth)
          and np.array_equal(t3.numpy(), wmx) and np.array_equal(t4.numpy(), wsm))
    q.put((rank, bool(ok)))
    d.barrier()
    d.destroy_process_group()


def test_segmentation_exchange_points_gloo():
    """The three exchange points of a segmentation call on the sharded map (ifx_owner_exchange(h, 200, ...)), with the encodings the
    library uses -- boxes with negated maxima under one int32 MIN, uint16 depth with disjoint supports summed as int32 words,
    per-instance MAX / SUM -- over gloo with two ranks: every rank ends with the unsharded values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_segworker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _camworker(rank, world, port, q):
    import torch

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w, d = ifd.init("gloo")
    rng = np.random.RandomState(11)
    root = 1
    # exchange 5 with a tracking rank (op 5 | root << 8): the prediction block -- one owner per pixel, zeros elsewhere -- SUMmed as int32 words to the ROOT only; the
    # 16-byte tail (vote mass) to everybody (op 1); exchange 310 (op 4 | root << 8): the 200-byte pose block broadcast from the root
    full = rng.standard_normal((3000, 8)).astype(np.float32)
    full[::53] = np.float32(-0.0); full[3::71] = np.nan
    owner = rng.randint(0, world, 3000)
    mine = np.where((owner == rank)[:, None], full, np.float32(0)).astype(np.float32)
    before = mine.view(np.int32).copy()
    t = torch.from_numpy(mine.view(np.int32).copy())
    d.reduce(t, dst=root, op=d.ReduceOp.SUM)
    tail = torch.tensor([100 + rank, 0, 0, 0], dtype=torch.int32)
    d.all_reduce(tail, op=d.ReduceOp.SUM)
    pose = torch.from_numpy(np.full(50, rank + 7, np.int32))
    d.broadcast(pose, src=root)
    q.put((rank, full.view(np.int32).copy(), before, t.numpy().copy(), tail.numpy().copy(), pose.numpy().copy()))
    d.barrier()
    d.destroy_process_group()


def test_camera_indexed_reduce_and_pose_broadcast_gloo():
    """K streams, camera k tracked by rank k (BASELINE configuration 5): the prediction of camera k's frame is reduced to rank k alone (ifx_owner_exchange op 5 | root << 8:
    ncclReduce in ifx_comm.hip, dist.reduce in the torch transport), its vote-mass tail all-reduced, the pose block broadcast from rank k (op 4 | root << 8) -- over gloo
    with two ranks: the root ends with the unsharded image bit for bit, every rank with the summed tail and the root's pose block."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_camworker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = 1
    assert np.array_equal(res[root][3], res[root][1])                   # the root holds the merged prediction
    for r_ in res:
        assert np.array_equal(r_[4], np.array([201, 0, 0, 0], np.int32))   # the tail reached everybody
        assert np.array_equal(r_[5], np.full(50, root + 7, np.int32))       # the root's pose block everywhere
