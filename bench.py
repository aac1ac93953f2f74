#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame dense surfel pipeline on MI355X (BASELINE.json metric).

One "step" = one 640x480 RGB-D frame through the whole hot path: preprocess -> 3-level ICP+RGB
tracking (SO(3) pre-alignment, 4/5/10 Gauss-Newton iterations) -> index map -> association/fusion
-> index map -> clean/append -> surfel-id render -> splat prediction + fill-in, plus the instance
layer at the reference's adaptive cadence (whetherDoSegmentation every frame; mask clean-up, vote
update and the label scan when it fires), into a pre-populated synthetic N-surfel map (default 5M).
Frames are resident in HBM before the timed region starts.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run
(one rank per GPU, RCCL).  Default: every rank runs the same workload on its own stream + map replica (weak scaling, no
data-path collective); `--sharded`: ONE stream into ONE map spatially sharded across the ranks, the exchanges of a frame
enqueued by libifx.so itself on the communicator it is handed.  Barrier + synchronize on both sides
of the timed region, max over ranks, rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


GN_ITERS = (10, 5, 4)     # Gauss-Newton iterations on pyramid levels 0, 1, 2
GN_PERSIST = 0            # ifx option gn_persist (a bit per level): levels whose iterations run in ONE persistent launch (k_gn_level); default since round 4: none


def level_avg(persist=None):
    """pixel count of the average two-launch Gauss-Newton iteration / P: the levels that do not use the persistent kernel"""
    persist = GN_PERSIST if persist is None else persist
    px = sum(GN_ITERS[l] / 4.0 ** l for l in range(3) if not persist & (1 << l))
    n = sum(GN_ITERS[l] for l in range(3) if not persist & (1 << l))
    return px / n if n else 0.0


def gn_level_bytes(P, persist=None):
    """average k_gn_level launch: per pixel 39 B of frame-side constants once (vertex + normal 24, depth 4, intensity 1 + 4x4 window ~4 after reuse, gradients 4) + per
    iteration 41 B of gathers (model vertex + normal 24, depth 4, intensity 1, cloud point 12), over the levels that use it"""
    persist = GN_PERSIST if persist is None else persist
    lv = [l for l in range(3) if persist & (1 << l)]
    return sum(P / 4.0 ** l * (39.0 + GN_ITERS[l] * 41.0) for l in lv) / len(lv) if lv else 0.0


def algorithmic_bytes(kernel: str, n_slots: int, P: int, active_fraction: float = 0.3, vl=None) -> float:
    """Algorithmic HBM bytes of ONE launch of `kernel`.  Deliberately conservative:
    only streams every launch must touch are counted (e.g. the key-image atomics of the rasteriser and the clean pass's window taps are not).
    vl = (entries of the time-window view list, stable entries outside it) of the run, from ifx_view_list_stats."""
    nw, no = (vl if vl else (0, 0))
    table = {
        # map: streaming culls over all slots
        "cull_frame": n_slots * 24.0,                       # the one scan of the store (view list): times 8 B + position/confidence 16 B of every slot
        "cull_raster": n_slots * 24.0,                      # pos+conf 16 B + times 8 B per slot
        "cull_clean": n_slots * (8.0 + 16.0 * active_fraction),   # times for every slot, position only inside the time window
        "index_project": n_slots * (8.0 + 16.0 * active_fraction),
        "count_colour": n_slots * (192.0 + 8 + 8 + 4),
        # map: list-driven passes (the frame path): per list entry the 4-B entry + the fields the pass reads of that slot
        "index_list": nw * (4.0 + 8 + 16),                  # entry, times, position + confidence
        "clean_view": nw * (4.0 + 8 + 16 + 16),             # + normal / radius of the candidates (window taps: L2-resident image, not counted)
        "raster_view": (nw + no) * (4.0 + 8 + 16 + 16),     # both lists: entry, times, position, normal / radius (key-image atomics not counted)
        "clean_raster_view": (nw + no) * (4.0 + 8 + 16 + 16),   # the fused walk (option clean_raster): the SAME records once for the clean and the raster (taps and atomics not counted)
        "fuse_update": (P / 4.0) * (4.0 + 96),              # upper bound: every active pixel matched (association word + 48 B read + 48 B written in place)
        "project_bbox": P * (4.0 + 192),                    # id image + the 12 vote planes of the surfel under every pixel
        "count_colour_px": P * (4.0 + 192 + 8),
        # tracker: per-pixel passes, averaged over the pyramid levels a launch can run at
        "gn_level": gn_level_bytes(P),
        "icp_residual": P * level_avg() * (48.0 + 22.0),     # ICP 24 B coalesced + 24 B gathered; residual 14 B read + 8 B written
        "rgb_step_solve": P * level_avg() * 24.0,            # 8-B record + 4 B gradients + 12 B gathered cloud point
        "bilateral_metric": P * (2.0 + 2 + 4 + 4),
        "splat_resolve": P * (8.0 + 16 + 16 + 4 + 4 + 2 + 16 + 16 + 4),
        "index_resolve": P * (8.0 + 4 + 48 + 16),
        "associate": P * (4.0 * 4 + 3 + 40),
    }
    return table.get(kernel, 0.0)


# which ms/frame column of the metric a kernel is billed to (roofline.stages)
STAGE_OF = {"track": ("icp_residual", "rgb_step_solve", "gn_level", "model_l0", "model_down"),
            "fuse": ("cull_frame", "cull_raster", "raster_list", "cull_clean", "clean_list", "clean_view", "raster_view", "clean_raster_view", "index_list", "index_project", "index_resolve", "index_resolve_taps",
                     "associate", "fuse_update", "splat_resolve", "tile_count", "tile_scan", "tile_fill", "tile_raster", "raster_finish", "new_flags_count", "append_scan"),
            "instance": ("count_colour", "count_colour_px", "project_bbox")}


def main():
    global GN_PERSIST
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--surfels", type=int, default=5_000_000)
    ap.add_argument("--loop", type=int, default=90, help="length of the closed camera loop (frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the bounded CPU-oracle sample (about 12 s of one core at 5M surfels)")
    ap.add_argument("--res", default="640x480", help="WxH of the synthetic stream (other BASELINE configurations; the metric is quoted at 640x480)")
    ap.add_argument("--sharded", action="store_true", help="one stream into ONE map spatially sharded across the ranks (owner = spatial hash of a surfel's position: every rank stores 1 / N of the "
                    "map; RCCL all-reduces of the key images and of the winners' attributes between the phases of a frame -- instancefusion_amd/sharded.py, DESIGN.md section 7; strong scaling; "
                    "segmentation calls through ifx_owner_segmentation_begin / _resume).  Default for --gpus N: one replica per rank")
    ap.add_argument("--sharded-projection", action="store_true", help="round 1's variant: every rank holds the whole map, the projection passes are sliced by slot range")
    ap.add_argument("--map-order", default="random", choices=["random", "morton"], help="order of the pre-populated synthetic map: `random` (default: surfels sampled uniformly, neighbours in the map are "
                    "unrelated in space -- the worst case for the passes that gather the visible part of the store) or `morton` (spatially coherent, as a map built frame by frame is)")
    ap.add_argument("--trace-steps", action="store_true", help="print the host time of every timed step to stderr (diagnostic)")
    ap.add_argument("--no-instance", action="store_true")
    ap.add_argument("--seg-host-frame", action="store_true", help="segmentation calls take the frame's RGB / depth from host memory (the reference's signature) instead of the resident frame")
    ap.add_argument("--pace", action="store_true", help="with --no-instance: wait for every frame's result before the next is enqueued (diagnostic)")
    ap.add_argument("--no-prefetch", action="store_true", help="no one-frame look-ahead (ifx_prefetch_frame_device)")
    ap.add_argument("--sharded-track-rows", type=int, default=0, help="the sharded-map leg shards the tracker's reductions over the ranks and all-reduces the 2 x 29 exact sums (option own_track_rows: 38 more collectives a frame; a measured loss at the BASELINE sizes, DESIGN.md section 7 -- the default keeps the tracker replicated)")
    ap.add_argument("--sharded-key-rs", type=int, default=0, help="the sharded-map leg runs the index-key exchanges as reduce-scatter + all-gather of the creation numbers (option own_key_rs: 12 instead of 16 bytes per key and link, two more collectives a frame; priced in DESIGN.md section 7: pays from about 1280x960 on)")
    ap.add_argument("--sharded-lazy-ids", type=int, default=1, help="the sharded-map leg exchanges the id keys of the sampled lattice only (option own_lazy_ids; 0: the whole id image every frame, 80 B per pixel)")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to ifx_set_option (experiments)")
    ap.add_argument("--close-loops", action="store_true", help="also run the local loop-closure detection every frame (the reference's closeLoops = true: predict() at the "
                    "tracked pose, INACTIVE prediction, model-to-model tracking, gates; thresholds of IF/map_interface/ElasticFusionInterface.cpp:43-45)")
    ap.add_argument("--fern-hook", action="store_true", help="with --close-loops: also the two read-backs per frame of the fern data base (findFrame inside the frame "
                    "through the fern callback, addFrame enqueued behind the frame and fetched in the next callback); the data base itself is host code (instancefusion_amd/host/ifx_ferns.hpp) and never matches here")
    ap.add_argument("--no-superpixels", action="store_true", help="skip the SLIC/merge/filter refinement of the masks (the reference always runs it)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the `value_sharded` leg (the same stream into ONE map spatially sharded over the ranks of this run, reported beside the replicas' `value`)")
    ap.add_argument("--config5-sets", type=int, default=6, help="frame sets (one frame per camera each) of the configuration-5 leg (N = 1: 8 cameras, inside the sharded leg; N > 1: N cameras, `value_config5`); 0 skips it")
    ap.add_argument("--config5-surfels", type=int, default=0, help="N > 1: surfels of the ONE shared map of the configuration-5 leg (0: --surfels x N, at most 50 M -- BASELINE configuration 5 at N = 8 within 20 %%)")
    ap.add_argument("--sharded-timeout", type=int, default=300, help="N > 1: seconds after which the line is printed without the sharded-map leg should that leg stall (0: wait for ever)")
    ap.add_argument("--extras-frames", type=int, default=60, help="frames of each extra leg at N = 1 (host entry point, closeLoops = true); 0 skips them")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one worker per GPU from this parent, which has not
        # touched HIP (a process that has initialised the GPU must never exec / re-launch), and relay rank 0's line
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to stdout when a communicator is created (torch's process group and
    # the library's own), so file descriptor 1 points at stderr from here on and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    from instancefusion_amd import dist as ifd

    rank, local_rank, world, dist = ifd.init("nccl")
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    import instancefusion_amd as ifx
    from instancefusion_amd import synth

    W, H = (int(v) for v in args.res.lower().split("x"))
    K = dict(fx=528.0 * W / 640, fy=528.0 * W / 640, cx=W / 2.0, cy=H / 2.0)
    P = W * H
    L = args.loop
    t_gen = time.time()
    one_map = args.sharded or args.sharded_projection
    srank = 0 if one_map else rank          # sharded: every rank is fed the same stream and map
    st = synth.make_stream(L, W, H, noise=True, loop_len=L, seed=synth.SEED + srank, **K)
    masks = [synth.canned_masks(st["obj"][i], st["scene"]) for i in range(L)]
    tick0 = 1000
    m = synth.make_map(args.surfels, st["scene"], st["poses_world"][0], tick0, seed=synth.SEED + 7 + srank, order=args.map_order)
    t_gen = time.time() - t_gen

    cap = args.surfels + 2_500_000
    owner = args.sharded        # at world == 1 too: a world of one on the sharded path (n_ranks = -1), every exchange point a one-rank RCCL collective
    ef = ifx.ElasticFusion(w=W, h=H, max_surfels=(cap // world + P + 500_000) if owner else cap, device=dev, **K, **(dict(n_ranks=(world if world > 1 else -1), rank=rank) if owner else {}))
    inst = ifx.InstanceFusion(ef)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ef.set_option(k_, int(v_))
        if k_ == "gn_persist":
            GN_PERSIST = int(v_)
    d_rgb = torch.from_numpy(st["rgb"]).cuda(dev)
    d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda(dev)
    torch.cuda.synchronize()

    # frame 0 initialises the tracker's previous-image pyramid; then the synthetic map replaces the
    # first-frame map and the model prediction is re-rendered from it
    osh = None
    if owner:
        from instancefusion_amd import sharded as ifsh

        osh = ifsh.OwnerShardedElasticFusion(ef, dist)   # hands libifx.so a RCCL communicator; from here on a frame is one library call
        ef.set_option("own_lazy_ids", int(bool(args.sharded_lazy_ids)))   # exchange 4 carries the id keys of the sampled lattice; the whole image with a segmentation call
        ef.set_option("own_key_rs", int(bool(args.sharded_key_rs)))
        ef.set_option("own_track_rows", int(bool(args.sharded_track_rows)))
        osh.process_frame_device(d_rgb[0].data_ptr(), d_dep[0].data_ptr())
        ef.upload(m)                                   # every rank is handed all rows and keeps the ones it owns
        ef.set_pose(st["poses"][0], tick0)
        osh.predict()
    else:
        ef.processFrame(st["rgb"][0], st["depth"][0])
        ef.upload(m)
        ef.set_pose(st["poses"][0], tick0)
        ef.combined_predict(st["poses"][0], tick0, tick0)
    want_sharded_leg = (not one_map) and (world > 1 or args.extras_frames > 0) and not args.close_loops and not args.no_sharded_leg
    if not want_sharded_leg:
        del m

    if args.close_loops:
        ef.set_loop_closure(True, 35000, 5e-5, 1e-5)
        fern_pending = [False]
        if args.fern_hook:
            def fern_cb(e):
                if fern_pending[0]:                 # Ferns::addFrame of the previous frame: enqueued behind it, complete by now
                    e.fern_frame_fetch()
                    fern_pending[0] = False
                e.fern_frame()                      # Ferns::findFrame's read-back of the prediction at the tracked pose
                return False
            ef.set_fern_callback(fern_cb)
    # ---- the instance layer inside the loop: whetherDoSegmentation every frame, a segmentation call when it fires (IF/main.cpp:108-307)
    seg = dict(frame=0, shift=0, last_true=None, calls=0, fast=0)

    def instance_step(i, allow=True):
        seg["frame"] += 1
        f = 100 + seg["frame"] + seg["shift"]
        if args.no_instance or not allow:
            if args.pace:
                ef.L.ifx_should_segment(ef.handle, -(1 << 30))   # (diagnostic) waits for the frame's result as the instance layer's decision does; never fires
            return
        if inst.whetherDoSegmentation(f):
            if seg["last_true"] is not None and f - seg["last_true"] <= 45:
                seg["fast"] += 1                    # the fast cadence (every 3rd frame) was chosen at least once
            seg["last_true"] = f
            mk, cl = masks[i]
            if mk.shape[0]:
                seg["calls"] += 1
                if osh is not None:   # sharded map: the owners' boxes / model depth / table statistics are merged at the call's exchange points
                    osh.process_segmentation(st["rgb"][i], st["depth"][i], mk, cl, seg["frame"], superpixels=not args.no_superpixels)
                else:
                    # the call belongs to the frame just processed, whose images are resident on the device (ifx_process_segmentation with rgb = depth = NULL);
                    # --seg-host-frame hands the host copies over instead, as InstanceFusion::ProcessSegmentation's signature has them (1.5 MB staged + uploaded)
                    if args.seg_host_frame:
                        inst.ProcessSegmentation(st["rgb"][i], st["depth"][i], mk, cl, seg["frame"], superpixels=not args.no_superpixels)
                    else:
                        inst.ProcessSegmentation(None, None, mk, cl, seg["frame"], superpixels=not args.no_superpixels)

    def place_call_in_window(n_frames):
        """The adaptive cadence (every 46th frame once the map carries votes, IF/Core/InstanceFusion.cpp:192-238) would leave a short
        timed window without any instance work.  The frame numbers handed to whetherDoSegmentation are shifted so that a call falls
        due in the MIDDLE of a window shorter than the cadence: such a window then holds one call (more instance work per frame than
        the steady state, never less); windows up to two cadences hold exactly one call, in the middle; longer windows keep the natural phase."""
        if args.no_instance or seg["last_true"] is None:
            return
        nxt = 100 + seg["frame"] + seg["shift"] + 1          # frame number the next step will present
        due = seg["last_true"] + 46
        if due > nxt + n_frames - 1 or due < nxt or n_frames < 2 * 46:   # (up to two cadences: ONE call, in the middle -- the natural phase gave such a window one call or two from run to run)
            seg["shift"] += due - (nxt + n_frames // 2)

    sh = osh
    if args.sharded_projection:
        from instancefusion_amd import sharded as ifsh

        sh = ifsh.ShardedElasticFusion(ef, rank, world, dist)

    def step(k, hint=True):
        i = k % L
        if sh is not None:
            if osh is not None and hint and not args.no_prefetch:   # the sharded path has the one-frame look-ahead too (frame side on the side stream, tracker parked behind the frame)
                ef.hint_next_frame_device(d_rgb[(k + 1) % L].data_ptr(), d_dep[(k + 1) % L].data_ptr())
            sh.process_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr())
            instance_step(i)
            return
        if hint and not args.no_prefetch:   # log replay: the next frame is known, its image-only work overlaps this frame's tracking
            ef.hint_next_frame_device(d_rgb[(k + 1) % L].data_ptr(), d_dep[(k + 1) % L].data_ptr())
        ef.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), k)
        if args.fern_hook and args.close_loops:
            if fern_pending[0]:
                ef.fern_frame_fetch()
            ef.fern_frame_async()                          # Ferns::addFrame's read-back of the end-of-frame prediction, fetched in the next callback
            fern_pending[0] = True
        instance_step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ef.sync()

    # The interpreter's cyclic garbage collector is not part of what is measured: with torch imported a full collection walks ~10^6 objects (45-50 ms: seventy
    # frames), and whether one falls into a timed window depends on the allocation history of everything before it (profiles/r04_t_*: one segmentation call of the
    # fast-cadence leg "took" 45 ms in runs without --trace-steps and 0.36 ms with it).  Collected once here, the survivors frozen, and switched off inside the timed
    # windows (a C++ host has no such pause; the C-ABI calls allocate nothing on the Python side that could form a cycle).
    import gc

    gc.collect()
    gc.freeze()

    def timed(k_first, n, fn):
        """n frames through fn(k), bracketed by barrier + synchronize; returns seconds (this rank)."""
        barrier()
        gc.disable()
        t0 = time.perf_counter()
        marks = []
        for k in range(k_first, k_first + n):
            fn(k)
            if args.trace_steps:
                marks.append((time.perf_counter(), ef.view_list_stats().get("scans", 0) if sh is None else 0))
        barrier()
        t1 = time.perf_counter()
        gc.enable()
        if args.trace_steps and rank == 0:   # host times of the steps (the host waits for every frame's result: they follow the device closely)
            ts = [m[0] for m in marks]
            sc = [m[1] for m in marks]
            print("step trace (ms; * = the step enqueued a view-list scan): " + " ".join(f"{(m - p_) * 1e3:.3f}{'*' if k_ and sc[k_] != sc[k_ - 1] else ''}" for k_, (m, p_) in enumerate(zip(ts, [t0] + ts[:-1])))
                  + f" | drain {(t1 - ts[-1]) * 1e3:.3f}", file=sys.stderr)
        return t1 - t0

    k = 1
    for _ in range(args.warmup):
        step(k); k += 1
    barrier()
    ef.stage_ms(reset=True)
    if osh is None:
        ef.superpixel_ahead_stats(reset=True)
    place_call_in_window(args.steps)
    seg["calls"] = 0
    k_timed = k
    if osh is not None:
        osh.exchange_stats(reset=True)
    dt = timed(k, args.steps, step); k += args.steps
    xstats = osh.exchange_stats() if osh is not None else None
    calls_in_window = seg["calls"]
    inst_ms = ef.stage_ms(reset=True)["instance"]          # the instance stage is always timed (two events per segmentation call)
    sp_ahead = ef.superpixel_ahead_stats(reset=True) if osh is None else None   # superpixels run ahead of the calls on the side stream (their device time is not in inst_ms)
    traj = ef.trajectory()                                  # poses up to the end of the timed region
    traj_all = traj
    dt = ifd.max_over_ranks(dt, dist, device=f"cuda:{dev}")
    # ms/frame split of the other stages: measured on the frames that follow, with the per-stage event records switched on
    # (eight marker packets per frame: they would cost ~4 % of the frame rate inside the timed region)
    n_split = 40
    ef.set_option("stage_timing", 1)
    for _ in range(n_split):
        step(k); k += 1
    ef.sync()
    stage = ef.stage_ms(reset=True)
    ef.set_option("stage_timing", 0)
    stage = {k_: v_ / n_split * args.steps for k_, v_ in stage.items()}     # scaled so that the common division below applies
    stage["instance"] = inst_ms
    n_live, n_slots = ef.count, ef.slots
    # trajectory error vs the synthetic ground truth over the timed frames (diagnostic)
    gt = np.stack([st["poses"][(k_timed + j) % L] for j in range(args.steps)])
    est = traj[-args.steps:]
    ate = float(np.sqrt(np.mean(np.sum((est[:, :3, 3] - gt[:, :3, 3]) ** 2, axis=1)))) if len(est) == args.steps else float("nan")

    # ---- roofline of the dominant kernel: per-launch HIP-event timing on the handle's stream
    roof = None
    if rank == 0:
        ef.set_option("kernel_timing", 1)
        ef.kernel_ms("__reset__")
        seg["calls"] = 0
        n_kt = 20
        place_call_in_window(n_kt)
        ef.sync()
        scans0 = ef.view_list_stats().get("scans", 0) if sh is None else 0
        for _ in range(n_kt):
            step(k); k += 1
        ef.sync()
        names = ["gn_level", "icp_residual", "rgb_step_solve", "so3_fused", "cull_frame", "cull_raster", "raster_list", "cull_clean", "clean_list", "clean_view", "raster_view", "clean_raster_view", "index_list", "index_project", "index_resolve",
                 "index_resolve_taps", "associate", "fuse_update", "bilateral_metric", "splat_resolve", "tile_count", "tile_scan", "tile_fill", "tile_raster", "raster_finish", "model_l0", "model_down", "new_flags_count",
                 "append_scan", "count_colour", "count_colour_px", "project_bbox"]
        # k_cull_frame is launched every frame and decides ON THE DEVICE whether the cached view list is still valid; a launch that finds it
        # valid returns at once.  Its algorithmic bytes are therefore the scan's bytes x (scans / launches) of this window.
        vls = ef.view_list_stats() if sh is None else {}
        scans_kt = (vls.get("scans", 0) - scans0) if sh is None else 0
        vl = (vls.get("window", 0), vls.get("outside", 0))
        alg = lambda n_: algorithmic_bytes(n_, n_slots, P, vl=vl)
        best, table = None, {}
        if ef.kernel_ms("gn_level")[1] == 0:   # (no level small enough for the persistent kernel at this resolution: every iteration is the two-launch form)
            GN_PERSIST = 0
        elif GN_PERSIST == 0:                  # (the library's own choice from 1280x960 on: the coarsest level)
            GN_PERSIST = 4
        for nme in names:
            avg, cnt = ef.kernel_ms(nme)
            table[nme] = dict(avg_ms=avg, launches=cnt, total_ms=avg * cnt)
            if cnt and alg(nme) > 0 and (best is None or avg * cnt > table[best]["total_ms"]):
                best = nme
        # the two launches of a Gauss-Newton iteration (icp_residual, rgb_step_solve) take the same time to within a per cent, so which of them
        # is "the" dominant kernel would flip from run to run: within 5 % the one that moves more algorithmic bytes is reported
        if best:
            for nme in names:
                t = table[nme]
                if t["launches"] and nme != best and t["total_ms"] > 0.95 * table[best]["total_ms"] and alg(nme) > alg(best):
                    best = nme
        scan_avg, scan_cnt = ef.kernel_ms("cull_frame@scan")
        ef.set_option("kernel_timing", 0)
        # HBM traffic per launch: PMC counters cannot be read from inside the process.  They are collected by tools/pmc_collect.sh (rocprofv3
        # --pmc FETCH_SIZE / WRITE_SIZE in separate passes over THIS command, corrected as MI355X_MICROARCH.md prescribes, per access pattern: tools/pmc_summary.py) into
        # profiles/<round>_pmc_traffic.json together with the slot count and the kernel list of that run.  A file is used only when it
        # describes this workload (same resolution, slot count within 10 %: tombstones come and go) and this kernel set; otherwise `traffic` is null -- never a stale constant.
        pmc, pmc_src = {}, None
        try:
            cands = sorted(f_ for f_ in os.listdir(os.path.join(ROOT, "profiles")) if f_.endswith("_pmc_traffic.json"))
            for f_ in reversed(cands):
                with open(os.path.join(ROOT, "profiles", f_)) as f:
                    j = json.load(f)
                meta = j.get("workload")
                if not meta or meta.get("res") != f"{W}x{H}" or abs(meta.get("surfel_slots", 0) - n_slots) > 0.10 * n_slots:
                    continue
                if not all(("k_" + n_) in j["kernels"] for n_ in names if table[n_]["launches"] and alg(n_) > 0 and n_ != "index_resolve_taps"):
                    continue
                pmc, pmc_src = j["kernels"], "profiles/" + f_
                break
        except (OSError, ValueError, KeyError):
            pass

        def entry(nme):
            b = alg(nme)
            if nme == "cull_frame" and table[nme]["launches"]:
                b *= min(1.0, scans_kt / table[nme]["launches"])
            ach = b / (table[nme]["avg_ms"] * 1e-3) / 1e9 if table[nme]["avg_ms"] > 0 else 0.0
            t = pmc.get("k_" + nme)
            traffic = (t["bytes_read"] + t["bytes_written"]) if t else None
            return dict(bound="hbm", kernel=nme, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=traffic, traffic_source=pmc_src if t else None, traffic_over_algorithmic=(round(traffic / b, 2) if (traffic and b > 0) else None),
                        fetch_factor=(t.get("fetch_factor") if t else None), avg_launch_ms=round(table[nme]["avg_ms"], 5), bytes_per_launch=b,
                        **({"scans": scans_kt, "launches": table[nme]["launches"]} if nme == "cull_frame" else {}))

        if best:
            roof = entry(best)
            roof["kernels"] = {k_: dict(avg_ms=round(v["avg_ms"], 5), launches=v["launches"]) for k_, v in table.items() if v["launches"]}
            # the dominant kernel by time is a latency-bound reduction; the passes that stream the whole surfel
            # store are reported next to it so that the bandwidth-bound part of the path has its roofline numbers too
            roof["streaming_passes"] = [entry(n_) for n_ in ("cull_frame", "cull_raster", "cull_clean", "index_project", "count_colour") if table.get(n_, {}).get("launches")]
            # ... and k_cull_frame's SCANNING launches on their own (most launches find the cached lists valid and return at once: the entry above averages over all of them)
            if scan_avg > 0 and scan_cnt:
                b_scan = alg("cull_frame")
                roof["streaming_passes"].append(dict(bound="hbm", kernel="cull_frame@scan", achieved=round(b_scan / (scan_avg * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                                                     frac=round(b_scan / (scan_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), avg_launch_ms=round(scan_avg, 5), bytes_per_launch=b_scan, launches=scan_cnt,
                                                     what="the launches of k_cull_frame that scanned the store (24 B per slot: position + confidence, times), by duration > 20 us"))
            # the list-driven passes that ARE the map stage on the frame path, and the per-call kernels of the instance layer: achieved fraction and waste ratio of each
            roof["map_passes"] = [entry(n_) for n_ in ("clean_raster_view", "raster_view", "clean_view", "index_list", "fuse_update", "associate", "index_resolve", "splat_resolve", "project_bbox", "count_colour_px")
                                  if table.get(n_, {}).get("launches")]
            # the two launches of a Gauss-Newton iteration PER PYRAMID LEVEL (the blended figure above averages 10 / 5 / 4 iterations at 1 : 1/4 : 1/16 of the pixels):
            # level 0 is where the bytes are, the coarse levels are pure launch latency
            lv = []
            for fam, bpp in (("icp_residual", 48.0 + 22.0), ("rgb_step_solve", 24.0)):
                for l in range(3):
                    avg, cnt = ef.kernel_ms(f"{fam}@L{l}")
                    if not cnt:
                        continue
                    b = P / 4.0 ** l * bpp
                    ach = b / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
                    lv.append(dict(kernel=f"{fam}@L{l}", level=l, pixels=int(P / 4 ** l), launches_per_frame=round(cnt / n_kt, 2), avg_launch_ms=round(avg, 5), bytes_per_launch=b, achieved=round(ach, 1),
                                   peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4)))
            roof["tracker_levels"] = lv
            # blended per stage: algorithmic bytes of the stage's launches per frame over the stage's measured ms per frame (the columns of the metric)
            stages = {}
            for sname, members in STAGE_OF.items():
                b_frame = 0.0
                for n_ in members:
                    if table.get(n_, {}).get("launches"):
                        b_l = alg("index_resolve" if n_ == "index_resolve_taps" else n_)
                        if n_ == "cull_frame":
                            b_l *= min(1.0, scans_kt / table[n_]["launches"])
                        b_frame += b_l * table[n_]["launches"] / n_kt
                ms = stage.get(sname, 0.0) / args.steps
                if sname == "instance":   # per segmentation call, not per frame
                    ms = (inst_ms / calls_in_window) if calls_in_window else 0.0
                    b_frame = b_frame * n_kt / max(1, seg["calls"])
                stages[sname] = dict(algorithmic_bytes=round(b_frame), ms=round(ms, 4), achieved=round(b_frame / (ms * 1e-3) / 1e9, 1) if ms > 0 else None, peak=HBM_PEAK_GBS, unit="GB/s",
                                     frac=round(b_frame / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None, per="segmentation call" if sname == "instance" else "frame")
            roof["stages"] = stages
            roof["view_list_entries"] = dict(window=vl[0], outside=vl[1])

    # ---- extra legs (N = 1 only, bounded): the same workload through the reference-shaped host entry point, and with closeLoops = true
    extras = {}
    if rank == 0 and world == 1 and sh is None and args.extras_frames > 0 and not args.close_loops:
        ne = args.extras_frames
        step(k, hint=False); k += 1                           # drains the one-frame look-ahead: the next leg hands over host buffers
        ef.sync()

        def host_step(kk):
            i = kk % L
            ef.processFrame(st["rgb"][i], st["depth"][i])     # ifx_process_frame: caller's host buffers, 1.54 MB H2D per frame, synchronous
            instance_step(i)

        for _ in range(5):
            host_step(k); k += 1
        place_call_in_window(ne)
        seg["calls"] = 0
        t_host = timed(k, ne, host_step); k += ne
        extras["value_host_entry"] = dict(value=round(ne / t_host, 2), unit="frames/s", frames=ne, segmentation_calls=seg["calls"],
                                          what="ifx_process_frame (ElasticFusion::processFrame's signature): host rgb/depth pointers, H2D inside the call, one host synchronisation per frame")
        # the same loop with option host_entry_async: the call returns when the frame's POSE is known (read back right behind the tracker); the frame's map passes finish under
        # whatever the caller does next.  `with_decision`: the caller asks whetherDoSegmentation after every frame (the reference's loop: it waits for the frame's result there,
        # so only the host copy of the next frame's depth image overlaps); `frames_only`: a caller that only feeds frames (a replay, a tracking-only client)
        ef.set_option("host_entry_async", 1)
        for _ in range(5):
            host_step(k); k += 1
        place_call_in_window(ne)
        seg["calls"] = 0
        t_ha = timed(k, ne, host_step); k += ne
        calls_ha = seg["calls"]

        def host_step_frames(kk):
            i = kk % L
            ef.processFrame(st["rgb"][i], st["depth"][i])

        for _ in range(3):
            host_step_frames(k); k += 1
        t_hf = timed(k, ne, host_step_frames); k += ne
        ef.sync()
        ef.set_option("host_entry_async", 0)
        extras["value_host_entry_async"] = dict(value=round(ne / t_ha, 2), unit="frames/s", frames=ne, segmentation_calls=calls_ha, frames_only=round(ne / t_hf, 2),
                                                what="ifx_process_frame with option host_entry_async (returns with the pose; map passes finish under the caller's next steps): `value` with "
                                                     "whetherDoSegmentation asked after every frame, `frames_only` without")
        # the reference-shaped entry with the NEXT frame announced from host memory before every call (ifx_hint_next_frame: what a log reader or a camera queue knows):
        # the announced frame's transfer and image-only work run on the side stream under the current frame and its tracker is parked behind it -- the resident path's
        # look-ahead plus 1.54 MB of host copy + H2D per frame; whetherDoSegmentation after every frame, calls on the resident frame
        def host_step_hinted(kk):
            i, n_ = kk % L, (kk + 1) % L
            ef.hint_next_frame(st["rgb"][n_], st["depth"][n_])
            ef.processFrame(st["rgb"][i], st["depth"][i])
            instance_step(i)

        for _ in range(5):
            host_step_hinted(k); k += 1
        place_call_in_window(ne)
        seg["calls"] = 0
        ef.lookahead_stats(reset=True)
        t_hh = timed(k, ne, host_step_hinted); k += ne
        la = ef.lookahead_stats()
        step(k, hint=False); k += 1                           # (consumes the last announcement's slot as an ordinary frame: the legs below start clean)
        ef.sync()
        extras["value_host_entry_hinted"] = dict(value=round(ne / t_hh, 2), unit="frames/s", frames=ne, segmentation_calls=seg["calls"], lookahead=la,
                                                 what="ifx_hint_next_frame(next frame, host pointers) + ifx_process_frame(current frame, host pointers) + whetherDoSegmentation, every frame")
        # the reference's own configuration: closeLoops = true (IF/map_interface/ElasticFusionInterface.cpp:43): predict() at the tracked pose,
        # INACTIVE prediction, model-to-model tracker and the gates on every frame; resident frames + look-ahead as in `value`
        ef.set_loop_closure(True, 35000, 5e-5, 1e-5)
        for _ in range(8):
            step(k); k += 1
        place_call_in_window(ne)
        seg["calls"] = 0
        t_lc = timed(k, ne, step); k += ne
        lcd = ef.loop_closure_diag()
        extras["value_close_loops"] = dict(value=round(ne / t_lc, 2), unit="frames/s", frames=ne, segmentation_calls=seg["calls"], candidates=int(lcd["candidates"]),
                                           what="as `value`, with the local loop-closure detection of closeLoops = true on every frame (no fern data base, no deformation applied)")
        step(k, hint=False); k += 1
        ef.sync()
        ef.set_loop_closure(False, 35000, 5e-5, 1e-5)

    # ---- the FAST cadence (IF/Core/InstanceFusion.cpp:194-233): on a young map -- less than 80 % of the sampled id image covered by stable surfels, hardly any votes --
    # whetherDoSegmentation fires every 3rd frame instead of every 46th.  The 5M-surfel map of `value` is old by construction, so this leg starts a map from nothing
    # and runs the same stream's first frames: frames/s and the instance stage per frame in that regime.
    if rank == 0 and world == 1 and sh is None and args.extras_frames > 0 and not args.close_loops and not args.no_instance:
        ef.sync()
        ef3 = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, device=dev, **K)
        inst3 = ifx.InstanceFusion(ef3)
        for kv in args.opt:
            k_, v_ = kv.split("=")
            ef3.set_option(k_, int(v_))
        nf3 = min(L, 90)
        fc = dict(calls=0, timed_calls=0, t0=None, first=12)

        trace3 = []

        def step3(i):
            ta_ = time.perf_counter()
            if i + 1 < nf3 and not args.no_prefetch:
                ef3.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
            ef3.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
            tb_ = time.perf_counter()
            fired = inst3.whetherDoSegmentation(100 + i)
            tc_ = time.perf_counter()
            if fired:
                mk, cl = masks[i]
                if mk.shape[0]:
                    inst3.ProcessSegmentation(None, None, mk, cl, i, superpixels=not args.no_superpixels)
                    fc["calls"] += 1
                    fc["timed_calls"] += 1 if i >= fc["first"] else 0
            trace3.append((i, int(fired), (tb_ - ta_) * 1e6, (tc_ - tb_) * 1e6, (time.perf_counter() - tc_) * 1e6))

        for i in range(fc["first"]):
            step3(i)
        ef3.sync(); torch.cuda.synchronize()
        ef3.stage_ms(reset=True)
        ef3.superpixel_ahead_stats(reset=True)
        gc.disable()
        t0 = time.perf_counter()
        for i in range(fc["first"], nf3):
            step3(i)
        t3a = time.perf_counter()
        ef3.sync()
        t3b = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter() - t0
        gc.enable()
        print(f"fast cadence leg: steps {(t3a - t0) * 1e3:.2f} ms, handle sync {(t3b - t3a) * 1e3:.2f} ms, device sync {(t0 + t3 - t3b) * 1e3:.2f} ms", file=sys.stderr)
        inst3_ms = ef3.stage_ms(reset=True)["instance"]
        sp3 = ef3.superpixel_ahead_stats(reset=True)
        n3 = nf3 - fc["first"]
        if args.trace_steps:   # host times of the leg's steps (us): enqueue of the frame | whetherDoSegmentation (waits for the frame's result) | the call
            for r_ in trace3[fc["first"]:]:
                print("fast cadence frame %2d fired %d  enqueue %6.0f  decide %6.0f  call %6.0f us" % r_, file=sys.stderr)
        n3 = nf3 - fc["first"]
        extras["value_fast_cadence"] = dict(value=round(n3 / t3, 2), unit="frames/s", frames=n3, segmentation_calls=fc["timed_calls"],
                                            frames_per_call=round(n3 / max(1, fc["timed_calls"]), 2), instance_ms_per_frame=round(inst3_ms / n3, 4),
                                            instance_ms_per_call=round(inst3_ms / max(1, fc["timed_calls"]), 4),
                                            superpixels_ahead=dict(runs=sp3["runs"], used_by_a_call=sp3["used"], ms_per_run_side_stream=round(sp3["ms"] / max(1, sp3["runs"]), 4)),
                                            surfels_live=ef3.count,
                                            host_us_median=dict(enqueue_frame=round(float(np.median([r_[2] for r_ in trace3[fc["first"]:]])), 1),
                                                                decide=round(float(np.median([r_[3] for r_ in trace3[fc["first"]:]])), 1),
                                                                call=round(float(np.median([r_[4] for r_ in trace3[fc["first"]:] if r_[1]] or [0.0])), 1)),
                                            what="the same stream into a map started from nothing: whetherDoSegmentation's fast cadence (a call every 3rd frame while less than 80 % of the "
                                                 "sampled id image is covered and the vote mass is low), resident frames + look-ahead as in `value`")
        ef3.close()

    # ---- the line rank 0 prints.  Everything but the two legs that follow (the sharded map beside the replicas, the CPU baseline) is known here; the handle-side
    # values are fetched now so that the builder touches no library state (the watchdog below may call it from a timer thread)
    vl_stats = ef.view_list_stats()
    x_ranks = osh.comm_ranks() if (osh is not None and xstats) else None
    lc_diag = ef.loop_closure_diag() if args.close_loops else None

    range_exceeded = ef.tracker_range_exceeded() if hasattr(ef, "tracker_range_exceeded") else None

    def build_line(sharded_leg_, cpu_):
        fps = (1 if one_map else world) * args.steps / dt
        out = {
            "metric": "frames/s + ms/frame (ICP|fuse|instance) at 640x480, 5M surfels, 1/2/4/8 GPU",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if one_map else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.surfels}-surfel map, {W}x{H} synthetic RGBD stream, 3-level ICP+RGB + surfel fuse + canned-mask instance votes{'' if args.no_superpixels else ' with superpixel refinement'}{' + local loop-closure detection' if args.close_loops else ''}",
                       "surfels_live": n_live, "surfel_slots": n_slots, "map_order": args.map_order, "parallelism": (f"spatially sharded map x{world}" if args.sharded else (f"sharded projection x{world}" if args.sharded_projection else f"replicas x{world}")), "loop_frames": L},
            "ms_per_frame_gpu": {k_: round(v / args.steps, 4) for k_, v in stage.items()},
            "instance": {"calls_in_window": calls_in_window, "ms_per_call": round(inst_ms / calls_in_window, 4) if calls_in_window else None,
                         "cadence_frames": 3 if seg["fast"] else 46, "ms_per_frame_at_cadence": round(inst_ms / calls_in_window / (3 if seg["fast"] else 46), 4) if calls_in_window else None,
                         "window_policy": "whetherDoSegmentation every frame; the cadence's phase is placed so that a window shorter than the cadence holds one call",
                         # ms_per_call is the main-stream span of a call (what the frame loop waits for); the superpixels of a call the cadence announced run ahead of it
                         # on the side stream, under the frame's tracker and map passes, and are timed there
                         **({"superpixels_ahead": dict(runs=sp_ahead["runs"], used_by_a_call=sp_ahead["used"], ms_per_run_side_stream=round(sp_ahead["ms"] / max(1, sp_ahead["runs"]), 4))} if sp_ahead else {})},
            **extras,
            **({"value_sharded": sharded_leg_} if sharded_leg_ else {}),
            "ate_rms_m": ate, "gen_s": round(t_gen, 1), "view_list": vl_stats,
            # run-time guard of the tracker's exact sums (ifx_tracker_range_exceeded): reductions of this whole run whose totals left the range in which the sums are order-independent
            # and equal the oracle's; 0 = every pose above is the pose of the fixed arithmetic
            "exact_sum_range_exceeded": range_exceeded,
            **({"exchange": {"transport": "RCCL collectives enqueued by libifx.so on the handle's stream (csrc/ifx_comm.hip)", "collectives_per_frame": round(xstats["collectives"] / args.steps, 2),
                             "bytes_per_frame_per_rank": round(xstats["bytes"] / args.steps), "bytes_per_pixel_per_frame": round(xstats["bytes"] / args.steps / P, 1), "rccl_ranks": x_ranks}} if xstats else {}),
            **({"loop_closure": {k_: (v_ if not isinstance(v_, np.ndarray) else None) for k_, v_ in lc_diag.items() if k_ != "est_pose"}} if args.close_loops else {}),
            "roofline": roof, "cpu_baseline": cpu_,
        }
        return out

    def tile_map(m_, n_target):
        """The benchmark map grown to n_target surfels without another minute of ray casting per rank: whole copies of it, each shifted by a fixed sub-centimetre offset
        (a denser sampling of the same surfaces: what make_map(n_target) would give), confidence / times / votes as they are."""
        n0 = m_["pc"].shape[0]
        reps = max(1, -(-n_target // n0))
        if reps == 1:
            return m_
        rng5 = np.random.RandomState(synth.SEED + 55)
        out_ = {}
        for key_, a_ in m_.items():
            out_[key_] = np.concatenate([a_] * reps)[:n_target]
        for r_ in range(1, reps):
            lo_, hi_ = r_ * n0, min((r_ + 1) * n0, n_target)
            out_["pc"][lo_:hi_, :3] += (rng5.uniform(-1, 1, 3) * 0.004).astype(np.float32)
        return out_

    def run_config5(K5, m5_, st5, d_rgb5, d_dep5):
        """K5 streams (K5 stretches of the same trajectory) into ONE map sharded over the `world` ranks of this run: camera contexts, camera c tracked by rank c % world ONLY
        (ifx_owner_set_tracking_rank: no tracker collective; its prediction is reduced to that rank, its pose block broadcast from it), and -- `ahead` -- the tracker of a
        camera's NEXT frame started on its rank's third stream the moment the camera's context is parked, under the other cameras' map phases (ifx_owner_track_ahead).
        Frames/s over all cameras with and without the run-ahead.  (tests/test_gpu_parity.py::test_config5_* hold the bit-parity of exactly this schedule, at 50M x 8 too.)"""
        from instancefusion_amd import sharded as ifsh5

        first5 = [((L - 12) // max(K5, 1)) * c_ for c_ in range(K5)] if K5 != 8 else [10 * c_ for c_ in range(8)]
        n_map = int(m5_["pc"].shape[0])
        res = {}
        for ahead5 in (0, 1):
            ef5 = ifx.ElasticFusion(w=W, h=H, max_surfels=n_map // world + P + 500_000, device=dev, **K, n_ranks=(world if world > 1 else -1), rank=rank)
            for kv in args.opt:
                k_, v_ = kv.split("=")
                ef5.set_option(k_, int(v_))
            osh5 = ifsh5.OwnerShardedElasticFusion(ef5, dist)
            ef5.camera_count(K5)
            osh5.process_frame_device(d_rgb5[0].data_ptr(), d_dep5[0].data_ptr())
            ef5.upload(m5_)
            ef5.set_pose(st5["poses"][0], tick0)
            osh5.predict()

            def set5(s_):
                for c_ in range(K5):
                    i_ = first5[c_] + 1 + s_
                    ef5.camera_select(c_)
                    ef5.owner_set_tracking_rank(c_ % world)
                    if c_ > 0 and s_ == 0:
                        ef5.owner_set_frame_pose(st5["poses"][i_].astype(np.float32))   # a camera enters with its extrinsic calibration, then tracks
                    if ahead5 and (s_ > 0 or c_ > 0):   # the camera whose frame was just processed is parked: its next frame's tracker starts now, on its rank
                        cp_, sp_ = (c_ - 1, s_) if c_ > 0 else (K5 - 1, s_ - 1)
                        j_ = first5[cp_] + 1 + sp_ + 1
                        if j_ < L:
                            ef5.owner_track_ahead(cp_, cp_ % world, d_rgb5[j_].data_ptr(), d_dep5[j_].data_ptr())
                    osh5.process_frame_device(d_rgb5[i_].data_ptr(), d_dep5[i_].data_ptr())

            def sync5():
                ef5.sync(); torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                    torch.cuda.synchronize()

            for s_ in range(2):
                set5(s_)
            sync5()
            served0 = ef5.owner_track_ahead(-1, 0)
            osh5.exchange_stats(reset=True)
            gc.disable()
            t0 = time.perf_counter()
            for s_ in range(2, 2 + args.config5_sets):
                set5(s_)
            sync5()
            t5 = ifd.max_over_ranks(time.perf_counter() - t0, dist, device=f"cuda:{dev}")
            gc.enable()
            xs5 = osh5.exchange_stats()
            nfr = K5 * args.config5_sets
            leg = dict(value=round(nfr / t5, 2), frames=nfr, trackers_served_ahead=ef5.owner_track_ahead(-1, 0) - served0)
            if world > 1:   # (the split and the exchange volume of the schedule that is reported as `value`)
                leg["exchange"] = dict(collectives_per_frame=round(xs5["collectives"] / nfr, 2), bytes_per_frame_per_rank=round(xs5["bytes"] / nfr), bytes_per_pixel_per_frame=round(xs5["bytes"] / nfr / P, 1))
                ef5.stage_ms(reset=True)
                ef5.set_option("stage_timing", 1)
                set5(2 + args.config5_sets)
                sync5()
                sm5 = ef5.stage_ms(reset=True)
                ef5.set_option("stage_timing", 0)
                leg["ms_per_frame_gpu_this_rank"] = {k_: round(v_ / K5, 4) for k_, v_ in sm5.items() if k_ != "instance"}
                leg["rccl_ranks"] = osh5.comm_ranks()
                leg["surfel_slots_this_rank"] = ef5.slots
            res["ahead" if ahead5 else "in_frame"] = leg
            ef5.close()
        out5 = dict(unit="frames/s", cameras=K5, n_ranks=world, surfels=n_map, **res)
        if world > 1:
            out5["value"] = res["ahead"]["value"]
            out5["scaling"] = "strong"
            out5["what"] = ("BASELINE configuration 5: %d concurrent %dx%d streams into ONE %d-surfel map spatially sharded over the %d ranks of this run (owner = spatial hash of a surfel's position), "
                            "camera k tracked by rank k only, the exchanges of every frame RCCL collectives enqueued by libifx.so (csrc/ifx_comm.hip): the prediction of camera k reduced to rank k, "
                            "its pose block broadcast from there.  `value` = frames/s over all cameras with every camera's next tracker running ahead on its rank under the other cameras' map "
                            "phases (`ahead`); `in_frame`: every frame tracks inside itself" % (K5, W, H, n_map, world))
        else:
            out5["what"] = ("BASELINE configuration 5 on one GPU: 8 streams into ONE sharded map (world of one: every exchange a one-rank RCCL collective, the camera-indexed "
                            "reduce included), camera contexts, time-sliced.  `in_frame`: every frame tracks inside itself.  `ahead`: every camera's next tracker runs on the "
                            "third stream as soon as the camera is parked (ifx_owner_track_ahead) and the frame commits its pose block -- the schedule of G ranks, where rank k "
                            "tracks camera k under the other cameras' map phases while the other ranks would otherwise wait for its pose; on ONE GPU that tracks all K cameras "
                            "the two chains (latency-bound tracker launches, map passes) share the device, and the run repeats the frame side (0.16 ms) it cannot hand over")
        return out5

    # The sharded leg at N > 1 is the one part of this benchmark that runs real multi-rank RCCL collectives enqueued by libifx.so.  Should it ever stall (a rank lost,
    # a communicator that does not come up), the replicas' measurement above must not be lost with it: after --sharded-timeout seconds rank 0 prints the line without
    # the leg (an "error" entry in its place) and every rank leaves.
    wd = None
    wd_state = {"leg": None}
    if want_sharded_leg and world > 1 and args.sharded_timeout > 0:
        import threading

        def _give_up():
            if rank == 0:
                part = wd_state["leg"]   # (the leg's own measurement, if the stall came behind it: in the timing of option own_track_rows, or in the configuration-5 leg)
                leg_ = dict(part, stalled_after=f"a later part of the leg did not finish within {args.sharded_timeout} s") if part else \
                    {"error": f"the sharded leg did not finish within {args.sharded_timeout} s; the line is reported without it", "n_ranks": world}
                os.write(json_fd, (json.dumps(build_line(leg_, None)) + "\n").encode())
            os._exit(3)   # (a stalled collective is not a success: the launcher records it; nothing is restarted or re-executed)

        wd = threading.Timer(args.sharded_timeout + (0 if rank == 0 else 5), _give_up)
        wd.daemon = True
        wd.start()

    # ---- the north star's partitioning beside the replicas: the SAME stream (rank 0's) into ONE map spatially sharded over the ranks of this run -- every rank stores
    # the surfels it owns, the exchanges of a frame are RCCL collectives enqueued by libifx.so.  At N = 1 a world of one: the fixed cost of the mode.
    sharded_leg = None
    if want_sharded_leg:
        try:   # (at N > 1 the one part that depends on multi-rank RCCL: a failure here is reported in the line, it does not take the replicas' measurement with it)
            from instancefusion_amd import sharded as ifsh

            ns = min(args.steps, max(args.extras_frames, 40)) if world == 1 else min(args.steps, 100)
            if world > 1:   # every rank is fed rank 0's stream and map
                st2 = synth.make_stream(L, W, H, noise=True, loop_len=L, seed=synth.SEED, **K) if rank != 0 else st
                m2 = synth.make_map(args.surfels, st2["scene"], st2["poses_world"][0], tick0, seed=synth.SEED + 7, order=args.map_order) if rank != 0 else m
                d_rgb2 = torch.from_numpy(st2["rgb"]).cuda(dev) if rank != 0 else d_rgb
                d_dep2 = torch.from_numpy(st2["depth"].view(np.int16)).cuda(dev) if rank != 0 else d_dep
                masks2 = [synth.canned_masks(st2["obj"][i], st2["scene"]) for i in range(L)] if rank != 0 else masks
            else:
                st2, m2, d_rgb2, d_dep2, masks2 = st, m, d_rgb, d_dep, masks
            ef.sync()
            ef2 = ifx.ElasticFusion(w=W, h=H, max_surfels=cap // world + P + 500_000, device=dev, **K, n_ranks=(world if world > 1 else -1), rank=rank)
            inst2 = ifx.InstanceFusion(ef2)
            for kv in args.opt:
                k_, v_ = kv.split("=")
                ef2.set_option(k_, int(v_))
            osh2 = ifsh.OwnerShardedElasticFusion(ef2, dist)
            ef2.set_option("own_lazy_ids", int(bool(args.sharded_lazy_ids)))
            ef2.set_option("own_key_rs", int(bool(args.sharded_key_rs)))
            ef2.set_option("own_track_rows", int(bool(args.sharded_track_rows)))
            osh2.process_frame_device(d_rgb2[0].data_ptr(), d_dep2[0].data_ptr())
            ef2.upload(m2)
            ef2.set_pose(st2["poses"][0], tick0)
            osh2.predict()
            m5 = m2 if args.config5_sets > 0 else None   # (kept for the configuration-5 leg below)
            del m2, m
            seg2 = dict(frame=0, calls=0)

            def step2(kk):
                i = kk % L
                if not args.no_prefetch:
                    ef2.hint_next_frame_device(d_rgb2[(kk + 1) % L].data_ptr(), d_dep2[(kk + 1) % L].data_ptr())
                osh2.process_frame_device(d_rgb2[i].data_ptr(), d_dep2[i].data_ptr())
                seg2["frame"] += 1
                if not args.no_instance and inst2.whetherDoSegmentation(100 + seg2["frame"]):
                    mk, cl = masks2[i]
                    if mk.shape[0]:
                        seg2["calls"] += 1
                        osh2.process_segmentation(st2["rgb"][i], st2["depth"][i], mk, cl, seg2["frame"], superpixels=not args.no_superpixels)

            def barrier2():
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                ef2.sync()

            k2 = 1
            for _ in range(12):
                step2(k2); k2 += 1
            barrier2()
            osh2.exchange_stats(reset=True)
            seg2["calls"] = 0
            gc.disable()
            t0 = time.perf_counter()
            for _ in range(ns):
                step2(k2); k2 += 1
            barrier2()
            dt2 = time.perf_counter() - t0
            gc.enable()
            xs2 = osh2.exchange_stats()
            dt2 = ifd.max_over_ranks(dt2, dist, device=f"cuda:{dev}")
            calls2 = seg2["calls"]
            ef2.stage_ms(reset=True)
            ef2.set_option("stage_timing", 1)
            n_split2 = 20
            for _ in range(n_split2):
                step2(k2); k2 += 1
            ef2.sync()
            stage2 = ef2.stage_ms(reset=True)
            ef2.set_option("stage_timing", 0)
            sharded_leg = dict(value=round(ns / dt2, 2), unit="frames/s", frames=ns, scaling="strong", n_ranks=world, rccl_ranks=osh2.comm_ranks(), segmentation_calls=calls2,
                               ms_per_frame_gpu={k_: round(v_ / n_split2, 4) for k_, v_ in stage2.items() if k_ != "instance"},
                               exchange={"transport": "RCCL collectives enqueued by libifx.so on the handle's stream (csrc/ifx_comm.hip)", "collectives_per_frame": round(xs2["collectives"] / ns, 2),
                                         "bytes_per_frame_per_rank": round(xs2["bytes"] / ns), "bytes_per_pixel_per_frame": round(xs2["bytes"] / ns / P, 1),
                                         "id_keys": "the sampled 10 x 10 lattice with every frame, the whole image with a segmentation call (option own_lazy_ids)" if args.sharded_lazy_ids else "the whole image with every frame",
                                         "index_keys": "reduce-scatter + all-gather of the creation numbers (option own_key_rs)" if args.sharded_key_rs else "all-reduce",
                                         "tracker": "reductions sharded over the ranks, 2 x 29 exact sums all-reduced per iteration (option own_track_rows)" if args.sharded_track_rows else "replicated (no collective)",
                                         "what": "all-reduce-equivalent bytes every rank hands to the collectives of a frame, segmentation calls of the window included: a ring moves 2 (N - 1) / N of them per link direction"},
                               surfel_slots_per_rank=ef2.slots, view_list=ef2.view_list_stats(),
                               what="ONE stream into ONE map spatially sharded over the ranks of this run (owner = spatial hash of a surfel's position; each rank stores its share); "
                                    "view lists + one-frame look-ahead as in `value`; `fuse` includes the exchanges")
            wd_state["leg"] = dict(sharded_leg)
            # the north star's tracker collective beside the replicated tracker, in the same leg: the reductions sharded over the ranks, the 2 x 29 exact sums all-reduced per
            # iteration (option own_track_rows) -- so that any multi-GPU run of this file prices it on hardware (DESIGN.md section 7 prices it on paper: 38 small collectives a frame)
            if not args.sharded_track_rows:
                try:
                    ef2.set_option("own_track_rows", 1)
                    for _ in range(6):
                        step2(k2); k2 += 1
                    barrier2()
                    osh2.exchange_stats(reset=True)
                    nr_ = min(ns, 40)
                    t0 = time.perf_counter()
                    for _ in range(nr_):
                        step2(k2); k2 += 1
                    barrier2()
                    dtr = ifd.max_over_ranks(time.perf_counter() - t0, dist, device=f"cuda:{dev}")
                    xsr = osh2.exchange_stats()
                    sharded_leg["tracker_rows"] = dict(value=round(nr_ / dtr, 2), unit="frames/s", frames=nr_, collectives_per_frame=round(xsr["collectives"] / nr_, 2),
                                                       what="the same leg with the tracker's reductions sharded over the ranks and the 2 x 29 exact sums all-reduced in f64 per iteration "
                                                            "(option own_track_rows; bit-identical poses); `value` of this leg keeps the tracker replicated")
                    ef2.set_option("own_track_rows", 0)
                except Exception as e_:   # (reported, not fatal: the leg's own measurement stands)
                    sharded_leg["tracker_rows"] = dict(error=str(e_)[:200])
            barrier2()
            ef2.close()

            # ---- BASELINE configuration 5: K streams into ONE shared map sharded over the ranks of this run (run_config5 below).  N = 1: 8 cameras in a world of one (what the
            # schedule costs and what the run-ahead is worth on one GPU); N > 1: N cameras, camera k tracked by rank k, the map of --config5-surfels over the N ranks through
            # the in-library RCCL exchanges -- `value_config5` of the line.  The SAME code either way.
            if m5 is not None:
                if world == 1:
                    sharded_leg["config5_world_of_one"] = run_config5(8, m5, st2, d_rgb2, d_dep2)
                else:
                    n5 = args.config5_surfels if args.config5_surfels > 0 else min(50_000_000, args.surfels * world)
                    extras["value_config5"] = run_config5(world, tile_map(m5, n5), st2, d_rgb2, d_dep2)
                del m5
        except Exception as e:   # noqa: BLE001
            import traceback

            traceback.print_exc()
            sharded_leg = {"error": f"{type(e).__name__}: {e}", "n_ranks": world}

    # ---- CPU baseline: the oracle (CPU restatement) on a bounded sample of the same workload, one core and all cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N = 1 only (the other ranks of a multi-GPU run would wait for it)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol

        n_cpu = min(args.surfels, 5_000_000)
        ncores = ol.usable_cores()                           # affinity capped by the cgroup quota (the GPU box shows 256 CPUs and grants 16)
        cpu_map = synth.make_map(n_cpu, st["scene"], st["poses_world"][0], tick0, seed=synth.SEED + 7)

        cpu_poses = {}

        def cpu_leg(threads, frames):
            ol.set_threads(threads)
            o = ol.Oracle(w=W, h=H, max_surfels=n_cpu + 1_000_000, **K)
            o.process_frame(st["rgb"][0], st["depth"][0])
            o.upload(cpu_map)
            o.set_pose(st["poses"][0], tick0)
            o.combined_predict(st["poses"][0], tick0, tick0)
            o.stage_ms(reset=True)
            tc = time.perf_counter()
            poses = []
            for kk in range(1, 1 + frames):
                poses.append(np.array(o.process_frame(st["rgb"][kk % L], st["depth"][kk % L]), np.float32).reshape(4, 4))
            t_frames = time.perf_counter() - tc
            cpu_poses[threads] = np.stack(poses)
            ms = o.stage_ms(reset=True)
            mk, cl = masks[frames % L]
            t_seg = 0.0
            if mk.shape[0] and not args.no_instance:          # one segmentation call (masks, superpixels, flood fill, votes, label scan)
                tc = time.perf_counter()
                o.process_segmentation(st["rgb"][frames % L], st["depth"][frames % L], mk, cl, 200, flags=0 if args.no_superpixels else 2)
                t_seg = time.perf_counter() - tc
            o.close()
            return dict(value=round(frames / t_frames, 4), unit="frames/s", cores=threads, frames=frames,
                        ms_per_frame={"track": round(ms["track"] / frames, 2), "fuse": round(ms["fuse"] / frames, 2)}, instance_ms_per_call=round(t_seg * 1e3, 1))

        one = cpu_leg(1, max(2, args.cpu_frames // 2))
        allc = cpu_leg(ncores, args.cpu_frames) if ncores > 1 else one
        cpu = dict(value=allc["value"], unit="frames/s", cores=allc["cores"], kind="port",
                   sample=f"{allc['frames']} frames of the same {W}x{H} stream into the same {n_cpu}-surfel synthetic map + one segmentation call (timed apart), OpenMP on all {allc['cores']} usable host cores: the tracker over image rows, the map renders over surfel ranges into per-thread z-buffers merged in draw order, association / stability tests over pixels / surfels (compaction and the append scan sequential); "
                          f"`one_core`: {one['frames']} frames on one core",
                   ms_per_frame=allc["ms_per_frame"], instance_ms_per_call=allc["instance_ms_per_call"], one_core=one)
        del cpu_map
        # parity inside the benchmark itself: the oracle's poses of frames 1.. (same stream, same map, same start) against the poses the GPU path produced for those
        # frames in THIS run -- resident frames, look-ahead, lazy compaction, cached view lists, segmentation calls in between (votes do not move poses)
        if n_cpu == args.surfels and sh is None and not args.close_loops:
            op = cpu_poses[allc["cores"]]
            gp = traj_all[1:1 + op.shape[0]]
            if gp.shape[0] == op.shape[0]:
                neq = int(sum(not np.array_equal(a_, b_) for a_, b_ in zip(gp, op)))
                cpu["parity_in_bench"] = dict(poses=int(op.shape[0]), not_bit_equal=neq, max_abs_diff=float(np.abs(gp - op).max()),
                                              what="the CPU oracle's poses of the first frames of this run against the GPU path's own poses of the same frames (bit equality of all 16 entries)")

    if wd is not None:
        wd.cancel()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(build_line(sharded_leg, cpu)) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
