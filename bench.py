#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame dense surfel pipeline on MI355X (BASELINE.json metric).

One "step" = one 640x480 RGB-D frame through the whole hot path: preprocess -> 3-level ICP+RGB
tracking (SO(3) pre-alignment, 4/5/10 Gauss-Newton iterations) -> index map -> association/fusion
-> index map -> clean/append -> surfel-id render -> splat prediction + fill-in, plus the instance
layer at the reference's adaptive cadence (whetherDoSegmentation every frame; mask clean-up, vote
update and the label scan when it fires), into a pre-populated synthetic N-surfel map (default 5M).
Frames are resident in HBM before the timed region starts.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run
(one rank per GPU, RCCL): every rank runs the same workload on its own stream + map replica
("replicas only" -- DESIGN.md section Multi-GPU), barrier + synchronize on both sides of the timed
region, max over ranks, rank 0 prints one JSON line.
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


LEVEL_AVG = (10 + 5 / 4.0 + 4 / 16.0) / 19.0   # pixel count of the average Gauss-Newton launch / P (10, 5, 4 iterations on levels 0, 1, 2)


def algorithmic_bytes(kernel: str, n_slots: int, P: int, active_fraction: float = 0.3) -> float:
    """Algorithmic HBM bytes of ONE launch of `kernel`.  Deliberately conservative:
    only streams every launch must touch are counted (e.g. the normal/radius reads of listed surfels are not)."""
    table = {
        # map: streaming culls over all slots
        "cull_raster": n_slots * 24.0,                      # pos+conf 16 B + times 8 B per slot
        "cull_clean": n_slots * (8.0 + 16.0 * active_fraction),   # times for every slot, position only inside the time window
        "index_project": n_slots * (8.0 + 16.0 * active_fraction),
        "count_colour": n_slots * (192.0 + 8 + 8 + 4),
        # tracker: per-pixel passes, averaged over the pyramid levels a launch can run at
        "icp_residual": P * LEVEL_AVG * (48.0 + 22.0),     # ICP 24 B coalesced + 24 B gathered; residual 14 B read + 8 B written
        "rgb_step_solve": P * LEVEL_AVG * 24.0,            # 8-B record + 4 B gradients + 12 B gathered cloud point
        "bilateral_metric": P * (2.0 + 2 + 4 + 4),
        "splat_resolve": P * (8.0 + 16 + 16 + 4 + 4 + 2 + 16 + 16 + 4),
        "index_resolve": P * (8.0 + 4 + 48 + 16),
        "associate": P * (4.0 * 4 + 3 + 40),
    }
    return table.get(kernel, 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--surfels", type=int, default=5_000_000)
    ap.add_argument("--loop", type=int, default=90, help="length of the closed camera loop (frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames of the bounded CPU-oracle sample (about 12 s of one core at 5M surfels)")
    ap.add_argument("--res", default="640x480", help="WxH of the synthetic stream (other BASELINE configurations; the metric is quoted at 640x480)")
    ap.add_argument("--sharded", action="store_true", help="one stream, every rank holds the map, projection passes sliced across ranks + RCCL all-reduce(MIN) of the key images "
                    "(instancefusion_amd/sharded.py; strong scaling; pays for tens of millions of surfels -- DESIGN.md section 7).  Default for --gpus N: one replica per rank")
    ap.add_argument("--no-instance", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true", help="no one-frame look-ahead (ifx_prefetch_frame_device)")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to ifx_set_option (experiments)")
    ap.add_argument("--close-loops", action="store_true", help="also run the local loop-closure detection every frame (the reference's closeLoops = true: predict() at the "
                    "tracked pose, INACTIVE prediction, model-to-model tracking, gates; thresholds of IF/map_interface/ElasticFusionInterface.cpp:43-45)")
    ap.add_argument("--fern-hook", action="store_true", help="with --close-loops: also the two read-backs per frame of the fern data base (findFrame inside the frame "
                    "through the fern callback, addFrame enqueued behind the frame and fetched in the next callback); the data base itself is host code (instancefusion_amd/host/ifx_ferns.hpp) and never matches here")
    ap.add_argument("--no-superpixels", action="store_true", help="skip the SLIC/merge/filter refinement of the masks (the reference always runs it)")
    args = ap.parse_args()

    import torch

    from instancefusion_amd import dist as ifd

    rank, local_rank, world, dist = ifd.init("nccl")
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    import instancefusion_amd as ifx
    from instancefusion_amd import synth

    W, H = (int(v) for v in args.res.lower().split("x"))
    K = dict(fx=528.0 * W / 640, fy=528.0 * W / 640, cx=W / 2.0, cy=H / 2.0)
    P = W * H
    L = args.loop
    t_gen = time.time()
    srank = 0 if args.sharded else rank          # sharded: every rank is fed the same stream and map
    st = synth.make_stream(L, W, H, noise=True, loop_len=L, seed=synth.SEED + srank, **K)
    masks = [synth.canned_masks(st["obj"][i], st["scene"]) for i in range(L)]
    tick0 = 1000
    m = synth.make_map(args.surfels, st["scene"], st["poses_world"][0], tick0, seed=synth.SEED + 7 + srank)
    t_gen = time.time() - t_gen

    cap = args.surfels + 2_500_000
    ef = ifx.ElasticFusion(w=W, h=H, max_surfels=cap, device=dev, **K)
    inst = ifx.InstanceFusion(ef)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ef.set_option(k_, int(v_))
    d_rgb = torch.from_numpy(st["rgb"]).cuda(dev)
    d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda(dev)
    torch.cuda.synchronize()

    # frame 0 initialises the tracker's previous-image pyramid; then the synthetic map replaces the
    # first-frame map and the model prediction is re-rendered from it
    ef.processFrame(st["rgb"][0], st["depth"][0])
    ef.upload(m)
    ef.set_pose(st["poses"][0], tick0)
    ef.combined_predict(st["poses"][0], tick0, tick0)
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
    frame_no = [0]
    sh = None
    if args.sharded:
        from instancefusion_amd import sharded as ifsh

        sh = ifsh.ShardedElasticFusion(ef, rank, world, dist)

    def step(k):
        i = k % L
        if sh is not None:
            sh.process_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr())
            frame_no[0] += 1
            if not args.no_instance and inst.whetherDoSegmentation(100 + frame_no[0]):
                mk, cl = masks[i]
                if mk.shape[0]:
                    inst.ProcessSegmentation(st["rgb"][i], st["depth"][i], mk, cl, frame_no[0], superpixels=not args.no_superpixels)
            return
        if not args.no_prefetch:   # log replay: the next frame is known, its image-only work overlaps this frame's tracking
            ef.hint_next_frame_device(d_rgb[(k + 1) % L].data_ptr(), d_dep[(k + 1) % L].data_ptr())
        ef.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), k)
        if args.fern_hook and args.close_loops:
            if fern_pending[0]:
                ef.fern_frame_fetch()
            ef.fern_frame_async()                          # Ferns::addFrame's read-back of the end-of-frame prediction, fetched in the next callback
            fern_pending[0] = True
        frame_no[0] += 1
        if not args.no_instance and inst.whetherDoSegmentation(100 + frame_no[0]):
            mk, cl = masks[i]
            if mk.shape[0]:
                inst.ProcessSegmentation(st["rgb"][i], st["depth"][i], mk, cl, frame_no[0], superpixels=not args.no_superpixels)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ef.sync()

    k0 = 1
    for k in range(k0, k0 + args.warmup):
        step(k)
    barrier()
    ef.stage_ms(reset=True)
    t0 = time.perf_counter()
    for k in range(k0 + args.warmup, k0 + args.warmup + args.steps):
        step(k)
    barrier()
    dt = time.perf_counter() - t0
    inst_ms = ef.stage_ms(reset=True)["instance"]          # the instance stage is always timed (two events per segmentation call)
    traj = ef.trajectory()                                  # poses up to the end of the timed region
    dt = ifd.max_over_ranks(dt, dist, device=f"cuda:{dev}")
    # ms/frame split of the other stages: measured on the frames that follow, with the per-stage event records switched on
    # (eight marker packets per frame: they would cost ~4 % of the frame rate inside the timed region)
    n_split = 40
    ef.set_option("stage_timing", 1)
    kk = k0 + args.warmup + args.steps
    for k in range(kk, kk + n_split):
        step(k)
    ef.sync()
    stage = ef.stage_ms(reset=True)
    ef.set_option("stage_timing", 0)
    stage = {k_: v_ / n_split * args.steps for k_, v_ in stage.items()}     # scaled so that the common division below applies
    stage["instance"] = inst_ms
    n_live, n_slots = ef.count, ef.slots
    # trajectory error vs the synthetic ground truth over the timed frames (diagnostic)
    gt = np.stack([st["poses"][(k0 + args.warmup + j) % L] for j in range(args.steps)])
    est = traj[-args.steps:]
    ate = float(np.sqrt(np.mean(np.sum((est[:, :3, 3] - gt[:, :3, 3]) ** 2, axis=1)))) if len(est) == args.steps else float("nan")

    # ---- roofline of the dominant kernel: per-launch HIP-event timing on the handle's stream
    roof = None
    if rank == 0:
        ef.set_option("kernel_timing", 1)
        ef.kernel_ms("__reset__")
        kk = k0 + args.warmup + args.steps + n_split
        for k in range(kk, kk + 20):
            step(k)
        ef.sync()
        names = ["icp_residual", "rgb_step_solve", "so3_fused", "cull_raster", "raster_list", "cull_clean", "clean_list", "index_project", "index_resolve", "associate",
                 "fuse_update", "bilateral_metric", "splat_resolve", "tile_count", "tile_scan", "tile_fill", "tile_raster", "raster_finish", "model_l0", "model_down", "new_flags_count", "append_scan", "count_colour"]
        best, table = None, {}
        for nme in names:
            avg, cnt = ef.kernel_ms(nme)
            table[nme] = dict(avg_ms=avg, launches=cnt, total_ms=avg * cnt)
            if cnt and algorithmic_bytes(nme, n_slots, P) > 0 and (best is None or avg * cnt > table[best]["total_ms"]):
                best = nme
        # the two launches of a Gauss-Newton iteration (icp_residual, rgb_step_solve) take the same time to within a per cent, so which of them
        # is "the" dominant kernel would flip from run to run: within 5 % the one that moves more algorithmic bytes is reported
        if best:
            for nme in names:
                t = table[nme]
                if t["launches"] and nme != best and t["total_ms"] > 0.95 * table[best]["total_ms"] and algorithmic_bytes(nme, n_slots, P) > algorithmic_bytes(best, n_slots, P):
                    best = nme
        ef.set_option("kernel_timing", 0)
        # HBM traffic per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
        # MI355X_MICROARCH.md prescribes for gfx950; collected offline on this same command, see profiles/README.md)
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_p_pmc_traffic.json")) as f:
                pmc = json.load(f)["kernels"]
        except (OSError, ValueError, KeyError):
            pass

        def entry(nme):
            b = algorithmic_bytes(nme, n_slots, P)
            ach = b / (table[nme]["avg_ms"] * 1e-3) / 1e9 if table[nme]["avg_ms"] > 0 else 0.0
            t = pmc.get("k_" + nme)
            return dict(bound="hbm", kernel=nme, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                        traffic=(t["bytes_read"] + t["bytes_written"]) if t else None, avg_launch_ms=round(table[nme]["avg_ms"], 5), bytes_per_launch=b)

        if best:
            roof = entry(best)
            roof["kernels"] = {k: dict(avg_ms=round(v["avg_ms"], 5), launches=v["launches"]) for k, v in table.items()}
            # the dominant kernel by time is a latency-bound reduction; the three streaming passes over the
            # whole surfel store are reported next to it so that the bandwidth-bound part of the path has its roofline numbers too
            roof["streaming_passes"] = [entry(n_) for n_ in ("cull_raster", "cull_clean", "index_project") if table.get(n_, {}).get("launches")]

    # ---- CPU baseline: the oracle (CPU restatement) on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N = 1 only (the other ranks of a multi-GPU run would wait for it)
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol

        n_cpu = min(args.surfels, 5_000_000)
        o = ol.Oracle(w=W, h=H, max_surfels=n_cpu + 1_000_000, **K)
        o.process_frame(st["rgb"][0], st["depth"][0])
        o.upload(synth.make_map(n_cpu, st["scene"], st["poses_world"][0], tick0, seed=synth.SEED + 7))
        o.set_pose(st["poses"][0], tick0)
        o.combined_predict(st["poses"][0], tick0, tick0)
        tc = time.perf_counter()
        for k in range(1, 1 + args.cpu_frames):
            o.process_frame(st["rgb"][k % L], st["depth"][k % L])
        tc = time.perf_counter() - tc
        cpu = dict(value=round(args.cpu_frames / tc, 4), unit="frames/s", cores=1, kind="port",
                   sample=f"{args.cpu_frames} frames of the same {W}x{H} stream into the same {n_cpu}-surfel synthetic map (no instance calls)")
        o.close()

    if rank == 0:
        fps = (1 if args.sharded else world) * args.steps / dt
        out = {
            "metric": "frames/s + ms/frame (ICP|fuse|instance) at 640x480, 5M surfels, 1/2/4/8 GPU",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if args.sharded else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic RGBD stream, 3-level ICP+RGB + surfel fuse + canned-mask instance votes{'' if args.no_superpixels else ' with superpixel refinement'}{' + local loop-closure detection' if args.close_loops else ''}, {args.surfels}-surfel map",
                       "surfels_live": n_live, "surfel_slots": n_slots, "parallelism": (f"sharded projection x{world}" if args.sharded else f"replicas x{world}"), "loop_frames": L},
            "ms_per_frame_gpu": {k: round(v / args.steps, 4) for k, v in stage.items()},
            "ate_rms_m": ate, "gen_s": round(t_gen, 1),
            **({"loop_closure": {k_: (v_ if not isinstance(v_, np.ndarray) else None) for k_, v_ in ef.loop_closure_diag().items() if k_ != "est_pose"}} if args.close_loops else {}),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
