#!/usr/bin/env python3
"""Time of the kNN colour smoothing (isflann) on a synthetic N-surfel map: python tools/knn_times.py [surfels]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(2, W, H, noise=True, **K)
m = synth.make_map(n, st["scene"], st["poses_world"][0], 1000)
ef = ifx.ElasticFusion(w=W, h=H, max_surfels=n + 100000, **K)
inst = ifx.InstanceFusion(ef)
ef.processFrame(st["rgb"][0], st["depth"][0]); ef.upload(m)
masks = np.zeros((1, H, W), np.uint8)
inst.ProcessSegmentation(st["rgb"][0], st["depth"][0], masks, np.array([1], np.int32), 0)      # label scan
for rep in range(3):
    t = time.perf_counter(); inst.flannKnnVoteSurfelMap(); dt = time.perf_counter() - t
    print(f"kNN smoothing of {n} surfels: {1000 * dt:.2f} ms")
ef.set_option("kernel_timing", 1); ef.kernel_ms("__reset__")
inst.flannKnnVoteSurfelMap(); ef.sync()
for k in ("knn_bounds", "knn_count", "scan_reduce", "scan_final", "knn_scatter", "knn_vote"):
    print(k, ef.kernel_ms(k))
