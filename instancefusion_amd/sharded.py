"""Multi-GPU drivers over libifx.so: the spatially sharded map (OwnerShardedElasticFusion: the collectives
are enqueued by the library itself, csrc/ifx_comm.hip) and, first in this file, round 1's sharded PROJECTION over full replicas.

Sharded projection:

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank holds the full surfel map and is
fed the same frames and masks, so the replicas stay bit-identical; the passes that stream the whole store with one
atomic per visible surfel -- the two index maps and the splat + id raster -- only handle this rank's slice of the slots
(`ifx_set_shard`), and between the four phases of a frame (`ifx_sharded_frame_phase`) the ranks combine their 64-bit key
images by an all-reduce(MIN).  min over disjoint slices == the global min (the key carries the slot id, so ties too),
hence the result equals the single-GPU run bit for bit (tests/test_gpu_parity.py::test_sharded_projection_emulated).

It pays for maps whose streaming passes dominate the frame (tens of millions of surfels); at the
BASELINE size of 5 M surfels replicas (bench.py's default for --gpus N) are faster.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

_SIGN = -(1 << 63)   # x ^ SIGN maps the unsigned order of the keys onto the signed order torch / RCCL reduce in


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface, also honoured on ROCm)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 3}


class KeyExchange:
    """The four key images of a handle as int64 torch tensors + the reduction across ranks."""

    def __init__(self, ef):
        import torch

        ptrs = [C.c_void_p() for _ in range(4)]
        n = C.c_int64()
        ef._chk(ef.L.ifx_key_images(ef.handle, *[C.byref(p) for p in ptrs], C.byref(n)), "ifx_key_images")
        self.t = [torch.as_tensor(_DevArray(p.value, n.value), device=f"cuda:{ef.cfgd['device']}") for p in ptrs]   # index, splat, ids, both

    def tensors(self, phase: int):
        return [self.t[0]] if phase in (0, 1) else self.t[1:]

    @staticmethod
    def reduce_min(tensors, dist):
        """Element-wise UNSIGNED minimum across the ranks of `dist`, in place."""
        for t in tensors:
            t.bitwise_xor_(_SIGN)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            t.bitwise_xor_(_SIGN)


class ShardedElasticFusion:
    """ElasticFusion.processFrame over `world` ranks.  `dist` is torch.distributed (initialised, backend nccl) or None for
    a single rank; `ef` an instancefusion_amd.ElasticFusion created on this rank's device."""

    def __init__(self, ef, rank: int, world: int, dist=None):
        import torch

        self.ef, self.rank, self.world, self.dist, self.torch = ef, rank, world, dist, torch
        ef._chk(ef.L.ifx_set_shard(ef.handle, rank, world), "ifx_set_shard")
        self.keys = KeyExchange(ef)
        main = C.c_void_p()
        ef._chk(ef.L.ifx_stream_handles(ef.handle, C.byref(main), None), "ifx_stream_handles")
        # the exchange is enqueued on the handle's own main stream (torch orders its RCCL stream against the current stream
        # with events), so a frame needs no host synchronisation between its phases
        self.stream = torch.cuda.ExternalStream(main.value, device=f"cuda:{ef.cfgd['device']}")

    def process_frame_device(self, d_rgb_ptr: int, d_depth_ptr: int):
        """One frame (device pointers, same content on every rank); returns nothing -- poses via ef.trajectory() / getCurrPose."""
        ef = self.ef
        for phase in range(4):
            ef._chk(ef.L.ifx_sharded_frame_phase(ef.handle, phase, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_sharded_frame_phase")
            if phase < 3 and self.dist is not None:
                with self.torch.cuda.stream(self.stream):
                    KeyExchange.reduce_min(self.keys.tensors(phase), self.dist)


def emulate_ranks(efs, d_rgb_ptr: int, d_depth_ptr: int, exchanges=None):
    """Test helper: `efs` are handles of ONE process (same GPU) configured as ranks 0..G-1 of G; the all-reduce is replaced by
    an element-wise minimum over their key images.  Exercises everything of the sharded mode except RCCL itself."""
    import torch

    xs = exchanges or [KeyExchange(e) for e in efs]
    for phase in range(4):
        for e in efs:
            e._chk(e.L.ifx_sharded_frame_phase(e.handle, phase, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_sharded_frame_phase")
        if phase < 3:
            for e in efs:
                e.sync()
            per_rank = [x.tensors(phase) for x in xs]
            for imgs in zip(*per_rank):
                m = imgs[0] ^ _SIGN
                for t in imgs[1:]:
                    m = torch.minimum(m, t ^ _SIGN)
                m ^= _SIGN
                for t in imgs:
                    t.copy_(m)
            torch.cuda.synchronize()
    return xs


# ---------------------------------------------------------------------------------------------------------------------------------
# Spatially sharded map (ifx_config.n_ranks > 1): every rank STORES 1 / G of the surfels (owner = spatial hash of the creation position).


def _exchange_spec(ef, phase):
    ptrs = (C.c_void_p * 8)()
    nbytes = (C.c_int64 * 8)()
    ops = (C.c_int32 * 8)()
    n = ef._chk(ef.L.ifx_owner_exchange(ef.handle, phase, ptrs, nbytes, ops, 8), "ifx_owner_exchange")
    return [(ptrs[k], nbytes[k], ops[k]) for k in range(n)]


class _DevWords:
    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 3}


class OwnerShardedElasticFusion:
    """ElasticFusion.processFrame over a spatially sharded map: `ef` was created with n_ranks = world, rank = this rank, on this rank's GPU
    (world of one: n_ranks = -1).  The collectives live INSIDE libifx.so (csrc/ifx_comm.hip): this class only hands the library a communicator --
    rank 0 draws a ncclUniqueId (ifx_comm_unique_id), `dist` (torch.distributed, any backend; None for a world of one) carries its 128 bytes to the
    other ranks, every rank calls ifx_owner_init_comm -- and a frame / predict / segmentation call / kNN smoothing is ONE library call each,
    with no return to Python between the phases.  `transport="torch"` keeps round 2's caller-driven exchanges (torch.distributed all-reduces on the
    handle's stream between ifx_owner_frame_phase calls) for hosts that bring their own transport."""

    def __init__(self, ef, dist=None, transport: str = "library"):
        import torch

        self.ef, self.dist, self.torch, self.transport = ef, dist, torch, transport
        self.dev = f"cuda:{ef.cfgd['device']}"
        if transport == "library":
            uid = np.zeros(128, np.uint8)
            multi = dist is not None and dist.get_world_size() > 1
            if not multi or dist.get_rank() == 0:
                ef._chk(ef.L.ifx_comm_unique_id(uid.ctypes.data_as(C.c_void_p)), "ifx_comm_unique_id")
            if multi:
                on_gpu = dist.get_backend() == "nccl"
                t = torch.from_numpy(uid).to(self.dev) if on_gpu else torch.from_numpy(uid)
                dist.broadcast(t, src=0)
                uid = t.cpu().numpy().copy()
            ef._chk(ef.L.ifx_owner_init_comm(ef.handle, uid.ctypes.data_as(C.c_void_p)), "ifx_owner_init_comm")
        else:
            main = C.c_void_p()
            ef._chk(ef.L.ifx_stream_handles(ef.handle, C.byref(main), None), "ifx_stream_handles")
            self.stream = torch.cuda.ExternalStream(main.value, device=self.dev)

    # ---- caller-driven exchanges (transport="torch")
    def _tensor(self, ptr, nbytes, op):
        u64 = op in (0, 6)
        return self.torch.as_tensor(_DevWords(ptr, nbytes // (8 if u64 else 4), "<i8" if u64 else "<i4"), device=self.dev)

    def _exchange(self, phase):
        spec = _exchange_spec(self.ef, phase)
        if spec and self.dist is not None:
            with self.torch.cuda.stream(self.stream):
                for ptr, nbytes, op in spec:
                    if (op & 0xFF) == 4:   # the pose block of the tracking rank (ifx_owner_set_tracking_rank): a broadcast from rank op >> 8, like ifx_comm_exchange's
                        self.dist.broadcast(self._tensor(ptr, nbytes, 1), src=op >> 8)
                        continue
                    if (op & 0xFF) == 5:   # int32 SUM to rank op >> 8 only (the prediction of a camera that one rank tracks)
                        self.dist.reduce(self._tensor(ptr, nbytes, 1), dst=op >> 8, op=self.dist.ReduceOp.SUM)
                        continue
                    t = self._tensor(ptr, nbytes, op)
                    if op in (0, 6):
                        KeyExchange.reduce_min([t], self.dist)
                        if op == 6:   # (option own_key_rs: the consumers want the creation numbers only -- this transport computes the whole MIN and strips the depth)
                            _strip_depth(t)
                    else:
                        self.dist.all_reduce(t, op={1: self.dist.ReduceOp.SUM, 2: self.dist.ReduceOp.MIN, 3: self.dist.ReduceOp.MAX}[op])

    # ---- the frame
    def process_frame_device(self, d_rgb_ptr: int, d_depth_ptr: int):
        ef = self.ef
        if self.transport == "library":
            ef._chk(ef.L.ifx_owner_process_frame_device(ef.handle, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr), 0), "ifx_owner_process_frame_device")
            return
        for phase in (310, 300, 301) + tuple(range(8)):   # (310: pose hand-over when one rank tracks; 300, 301: loop-closure renders when enabled;) 0..5 with their exchanges, 6 fill-in / dense flag, 7 publishes the frame result
            ef._chk(ef.L.ifx_owner_frame_phase(ef.handle, phase, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_owner_frame_phase")
            self._exchange(phase)

    def process_frame(self, rgb, depth):
        """ElasticFusion::processFrame's shape: host images in, currPose out (one synchronisation)."""
        ef = self.ef
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
        pose = np.zeros(16, np.float32)
        ef._chk(ef.L.ifx_owner_process_frame(ef.handle, rgb.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), 0, pose.ctypes.data_as(C.c_void_p)), "ifx_owner_process_frame")
        return pose.reshape(4, 4)

    def predict(self):
        """ElasticFusion::predict outside a frame (after upload / set_pose)."""
        ef = self.ef
        if self.transport == "library":
            ef._chk(ef.L.ifx_owner_predict(ef.handle), "ifx_owner_predict")
            return
        for step in range(3):
            ef._chk(ef.L.ifx_owner_predict_phase(ef.handle, step), "ifx_owner_predict_phase")
            if step < 2:
                self._exchange(4 + step)

    def exchange_stats(self, reset=False):
        out = np.zeros(2, np.int64)
        self.ef._chk(self.ef.L.ifx_owner_exchange_stats(self.ef.handle, out.ctypes.data_as(C.c_void_p), int(reset)), "ifx_owner_exchange_stats")
        return dict(collectives=int(out[0]), bytes=int(out[1]))

    def comm_ranks(self):
        """ranks of the library's communicator as RCCL counts them (ncclCommCount); 0 with transport="torch" """
        return self.ef._chk(self.ef.L.ifx_owner_comm_ranks(self.ef.handle), "ifx_owner_comm_ranks")

    def knn_vote_colour(self):
        """InstanceFusion::flannKnnVoteSurfelMap on the sharded map: all-gather of every rank's slots (20 B each), exact 10-NN of the owned surfels."""
        if self.transport == "library":
            self.ef._chk(self.ef.L.ifx_owner_knn_vote_colour(self.ef.handle), "ifx_owner_knn_vote_colour")
            return
        torch, dist = self.torch, self.dist
        pts, lab = _knn_export(self.ef, torch, self.dev)
        if dist is None:
            allp, alll, off = pts, lab, 0
        else:
            world, rank = dist.get_world_size(), dist.get_rank()
            n = torch.tensor([pts.shape[0]], dtype=torch.int64, device=self.dev)
            ns = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(ns, n)
            ns = [int(x.item()) for x in ns]
            nmax = max(ns)
            padp = torch.full((nmax, 4), float("nan"), dtype=torch.float32, device=self.dev); padp[:pts.shape[0]] = pts
            padl = torch.full((nmax,), -1, dtype=torch.int32, device=self.dev); padl[:lab.shape[0]] = lab
            gp = [torch.empty_like(padp) for _ in range(world)]; gl = [torch.empty_like(padl) for _ in range(world)]
            dist.all_gather(gp, padp); dist.all_gather(gl, padl)
            allp = torch.cat([gp[r][:ns[r]] for r in range(world)]).contiguous()
            alll = torch.cat([gl[r][:ns[r]] for r in range(world)]).contiguous()
            off = sum(ns[:rank])
        torch.cuda.synchronize()
        self.ef._chk(self.ef.L.ifx_owner_knn_vote(self.ef.handle, C.c_void_p(allp.data_ptr()), C.c_void_p(alll.data_ptr()), int(allp.shape[0]), int(off)), "ifx_owner_knn_vote")

    def ensure_ids(self):
        """Option own_lazy_ids: the whole id image (a frame exchanges the sampled lattice only).  With the library's communicator whoever reads the image completes it in
        place; a caller-driven transport does it here -- every rank together, before ef.image("ids_after") / camera_select."""
        if self.transport == "library":
            return
        if self.ef._chk(self.ef.L.ifx_owner_ids_begin(self.ef.handle), "ifx_owner_ids_begin") == 1:
            self._exchange(200)
            self.ef._chk(self.ef.L.ifx_owner_ids_resume(self.ef.handle), "ifx_owner_ids_resume")

    def process_segmentation(self, rgb, depth, masks, class_ids, frame: int, superpixels: bool = True, knn: bool = False):
        """InstanceFusion::processInstance on the sharded map (same masks on every rank): the owners' partial boxes, model depth and -- when the
        instance table overflows -- eviction statistics are merged at the call's exchange points; labels of the owned surfels: ef.labels()."""
        if self.transport == "library":
            ef = self.ef
            masks = np.ascontiguousarray(masks, np.uint8); cls = np.ascontiguousarray(class_ids, np.int32)
            rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
            ef._chk(ef.L.ifx_owner_process_segmentation(ef.handle, rgb.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), masks.ctypes.data_as(C.c_void_p),
                                                        cls.ctypes.data_as(C.c_void_p), int(masks.shape[0]), int(frame), (2 if superpixels else 0) | (1 if knn else 0)),
                    "ifx_owner_process_segmentation")
            return
        r = _seg_begin(self.ef, rgb, depth, masks, class_ids, frame, superpixels)
        while r == 1:
            self._exchange(200)
            r = self.ef._chk(self.ef.L.ifx_owner_segmentation_resume(self.ef.handle), "ifx_owner_segmentation_resume")
        if knn:
            self.knn_vote_colour()


def _knn_export(ef, torch, dev):
    pp, pl, n = C.c_void_p(), C.c_void_p(), C.c_int()
    ef._chk(ef.L.ifx_owner_knn_export(ef.handle, C.byref(pp), C.byref(pl), C.byref(n)), "ifx_owner_knn_export")
    if n.value == 0:
        return torch.zeros((0, 4), dtype=torch.float32, device=dev), torch.zeros((0,), dtype=torch.int32, device=dev)
    pts = torch.as_tensor(_DevWords(pp.value, n.value * 4, "<f4"), device=dev).view(-1, 4)
    lab = torch.as_tensor(_DevWords(pl.value, n.value, "<i4"), device=dev)
    return pts, lab


def _seg_begin(ef, rgb, depth, masks, class_ids, frame, superpixels):
    masks = np.ascontiguousarray(masks, np.uint8)
    cls = np.ascontiguousarray(class_ids, np.int32)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    depth = np.ascontiguousarray(depth, np.uint16)
    return ef._chk(ef.L.ifx_owner_segmentation_begin(ef.handle, rgb.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), masks.ctypes.data_as(C.c_void_p),
                                                     cls.ctypes.data_as(C.c_void_p), int(masks.shape[0]), int(frame), 2 if superpixels else 0), "ifx_owner_segmentation_begin")


def _strip_depth(t):
    """op 6 of ifx_owner_exchange: keys come back as (uint64) creation number, empty keys (all ones) whole."""
    low = t & 0xFFFFFFFF
    t.copy_(low.masked_fill(low == 0xFFFFFFFF, -1))


def _reduce_by_hand(efs, specs):
    import torch

    dev = f"cuda:{efs[0].cfgd['device']}"
    for e in efs:
        e.sync()
    for k in range(len(specs[0])):
        op, nbytes = specs[0][k][2], specs[0][k][1]
        u64 = op in (0, 6)
        ts = [torch.as_tensor(_DevWords(sp[k][0], nbytes // (8 if u64 else 4), "<i8" if u64 else "<i4"), device=dev) for sp in specs]
        if (op & 0xFF) == 5:                       # int32 SUM to rank op >> 8 only: the others keep what they hold (their own partial sums, never read)
            m = ts[0].clone()
            for t in ts[1:]:
                m += t
            ts[op >> 8].copy_(m)
            continue
        if (op & 0xFF) == 4:                       # broadcast from rank op >> 8
            m = ts[op >> 8].clone()
        elif u64:
            m = ts[0] ^ _SIGN
            for t in ts[1:]:
                m = torch.minimum(m, t ^ _SIGN)
            m ^= _SIGN
            if op == 6:
                _strip_depth(m)
        elif op == 1:
            m = ts[0].clone()
            for t in ts[1:]:
                m += t
        else:
            m = ts[0].clone()
            for t in ts[1:]:
                m = torch.minimum(m, t) if op == 2 else torch.maximum(m, t)
        for t in ts:
            t.copy_(m)
    torch.cuda.synchronize()


def emulate_owner_predict(efs):
    """ElasticFusion::predict of a sharded map outside a frame (after upload / set_pose), the exchanges done by hand."""
    for step in range(3):
        for e in efs:
            e._chk(e.L.ifx_owner_predict_phase(e.handle, step), "ifx_owner_predict_phase")
        if step < 2:
            _reduce_by_hand(efs, [_exchange_spec(e, 4 + step) for e in efs])


def emulate_owner_ranks(efs, d_rgb_ptr: int, d_depth_ptr: int):
    """Test helper: `efs` = handles of ONE process created with n_ranks = len(efs), rank = 0..G-1; the all-reduces are done by hand
    (element-wise unsigned minimum / int32 sum over the handles' buffers).  Everything of the sharded map except RCCL itself."""
    for phase in (310, 300, 301) + tuple(range(8)):   # 310: hand-over of the pose when one rank tracks; 300, 301: the renders of the loop-closure detection (no-ops unless enabled and due)
        for e in efs:
            e._chk(e.L.ifx_owner_frame_phase(e.handle, phase, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr)), "ifx_owner_frame_phase")
        specs = [_exchange_spec(e, phase) for e in efs]
        if specs[0]:
            _reduce_by_hand(efs, specs)


def emulate_owner_segmentation(efs, rgb, depth, masks, class_ids, frame: int, superpixels: bool = True):
    """A segmentation call on the handles of emulate_owner_ranks, in lock step, the exchanges done by hand."""
    rs = [_seg_begin(e, rgb, depth, masks, class_ids, frame, superpixels) for e in efs]
    while rs[0] == 1:
        assert all(r == 1 for r in rs), "the ranks disagree about the call's exchange points"
        _reduce_by_hand(efs, [_exchange_spec(e, 200) for e in efs])
        rs = [e._chk(e.L.ifx_owner_segmentation_resume(e.handle), "ifx_owner_segmentation_resume") for e in efs]
    assert all(r == 0 for r in rs)


def emulate_owner_ids(efs):
    """Option own_lazy_ids on the handles of emulate_owner_ranks: the whole id image (every shard's id render, the keys MIN-reduced by hand)."""
    rs = [e._chk(e.L.ifx_owner_ids_begin(e.handle), "ifx_owner_ids_begin") for e in efs]
    assert all(r == rs[0] for r in rs), "the ranks disagree about the state of the id image"
    if rs[0] == 1:
        _reduce_by_hand(efs, [_exchange_spec(e, 200) for e in efs])
        for e in efs:
            e._chk(e.L.ifx_owner_ids_resume(e.handle), "ifx_owner_ids_resume")


def emulate_owner_knn(efs):
    """The kNN smoothing on the handles of emulate_owner_ranks: the all-gather is a concatenation of the exports in rank order."""
    import torch

    dev = f"cuda:{efs[0].cfgd['device']}"
    ex = [_knn_export(e, torch, dev) for e in efs]
    allp = torch.cat([p for p, _ in ex]).contiguous()
    alll = torch.cat([l for _, l in ex]).contiguous()
    torch.cuda.synchronize()
    off = 0
    for e, (p, _) in zip(efs, ex):
        e._chk(e.L.ifx_owner_knn_vote(e.handle, C.c_void_p(allp.data_ptr()), C.c_void_p(alll.data_ptr()), int(allp.shape[0]), int(off)), "ifx_owner_knn_vote")
        off += int(p.shape[0])
