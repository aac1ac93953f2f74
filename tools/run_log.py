#!/usr/bin/env python3
"""Replays a recorded RGB-D log through the MI355X path -- the loop of the reference's main program
(IF/main.cpp:108-307) without its GUI: read frame -> ProcessFrame -> whetherDoSegmentation -> ProcessSegmentation with
replayed masks -> every `--flann-every` frames the kNN smoothing -> at the end the trajectory (.freiburg) and the two
PLY models, as ElasticFusion's destructor writes them (EF/ElasticFusion.cpp:99-128, 796-990).

    python tools/run_log.py LOG.klg --width 640 --height 480 --fx 528 --fy 528 --cx 320 --cy 240 \
        [--masks DIR] [--out PREFIX] [--max-frames N] [--no-instance]

LOG may be a `.klg` file (IF/utilities/RawLogReader.cpp) or a `data.txt` image list (IF/utilities/PNGLogReader.cpp).
Masks replace the Mask-RCNN bridge (BASELINE.json: "the Mask-RCNN call left as a stub that replays pre-computed masks"):
DIR/<frame index, 6 digits>.npz with `masks` (n x H x W uint8, 255 inside, sorted by area descending as
REF/build/mask_ori.py:117 delivers them) and `class_ids` (n int32); frames without a file get no segmentation.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--fx", type=float, default=528.0)
    ap.add_argument("--fy", type=float, default=528.0)
    ap.add_argument("--cx", type=float, default=320.0)
    ap.add_argument("--cy", type=float, default=240.0)
    ap.add_argument("--max-surfels", type=int, default=6_000_000)
    ap.add_argument("--masks", default=None)
    ap.add_argument("--out", default="ResultModel")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--no-instance", action="store_true")
    ap.add_argument("--flann-every", type=int, default=40, help="kNN smoothing when a segmentation happens more than this many frames after the last one (IF/main.cpp:36)")
    ap.add_argument("--flip-colors", action="store_true")
    ap.add_argument("--confidence", type=float, default=10.0, help="surfel stability threshold (the reference's constructor argument)")
    ap.add_argument("--no-close-loops", action="store_true", help="switch the local loop-closure detection off (the reference runs with closeLoops = true)")
    ap.add_argument("--labels", default=None, help="write bestIDInEachSurfel of the live surfels (int32, map order) to this file")
    args = ap.parse_args(argv)

    import torch  # noqa: F401  (binds the HIP runtime first)

    import instancefusion_amd as ifx
    from instancefusion_amd import logio

    if args.log.endswith(".txt"):
        reader = logio.PNGLogReader(args.log, args.width, args.height)
    else:
        reader = logio.RawLogReader(args.log, args.width, args.height, flipColors=args.flip_colors)
    ef = ifx.ElasticFusion(w=args.width, h=args.height, fx=args.fx, fy=args.fy, cx=args.cx, cy=args.cy, max_surfels=args.max_surfels, confidence=args.confidence)
    if not args.no_close_loops:
        ef.set_loop_closure(True, 35000, 5e-5, 1e-5)       # IF/map_interface/ElasticFusionInterface.cpp:43-45
    inst = ifx.InstanceFusion(ef)
    stamps, poses = [], []
    frame, last_flann, n_seg = 0, -1, 0     # lastTimeFlann = -1, IF/main.cpp:111
    t0 = time.perf_counter()
    while reader.hasMore() and (args.max_frames <= 0 or frame < args.max_frames):
        reader.getNext()
        pose = ef.processFrame(reader.rgb, reader.depth, timestamp=reader.timestamp)
        stamps.append(reader.timestamp)
        poses.append(pose)
        if not args.no_instance and args.masks and inst.whetherDoSegmentation(frame):
            f = os.path.join(args.masks, f"{frame:06d}.npz")
            if os.path.exists(f):
                d = np.load(f)
                flann = frame - last_flann > args.flann_every
                inst.ProcessSegmentation(reader.rgb, reader.depth, d["masks"], d["class_ids"], frame, isflann=flann, superpixels=True)
                last_flann = frame if flann else last_flann
                n_seg += 1
        frame += 1
    ef.sync()
    dt = time.perf_counter() - t0
    logio.save_freiburg(args.out + ".freiburg", stamps, poses)
    m = ef.download()
    n_geo = logio.save_ply(args.out + ".ply", m, confidence=args.confidence)
    n_ins = logio.save_ply(args.out + "_Instance.ply", m, confidence=args.confidence, instance=True)
    if args.labels:
        inst.labels().astype(np.int32).tofile(args.labels)
    print(f"{frame} frames in {dt:.2f} s ({frame / max(dt, 1e-9):.1f} frames/s incl. log decoding), {n_seg} segmentation calls, {ef.count} surfels, "
          f"{n_geo} stable surfels -> {args.out}.ply / _Instance.ply ({n_ins}), trajectory -> {args.out}.freiburg, "
          f"{ef.loop_closure_diag()['candidates'] if not args.no_close_loops else 0} local loop-closure candidates")
    ef.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
