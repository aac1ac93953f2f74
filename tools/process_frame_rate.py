#!/usr/bin/env python3
"""Frames/s of the synchronous host-buffer entry point (ifx_process_frame = the reference's ProcessFrame signature), PCIe included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
W, H, L = 640, 480, 90
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(L, W, H, noise=True, loop_len=L, **K)
m = synth.make_map(n, st["scene"], st["poses_world"][0], 1000)
ef = ifx.ElasticFusion(w=W, h=H, max_surfels=n + 2_500_000, **K)
ef.processFrame(st["rgb"][0], st["depth"][0]); ef.upload(m); ef.set_pose(st["poses"][0], 1000); ef.combined_predict(st["poses"][0], 1000, 1000)
for k in range(1, 31): ef.processFrame(st["rgb"][k % L], st["depth"][k % L])
t = time.perf_counter()
N = 150
for k in range(31, 31 + N): ef.processFrame(st["rgb"][k % L], st["depth"][k % L])
dt = time.perf_counter() - t
print(f"ifx_process_frame (host buffers, synchronous): {N / dt:.1f} frames/s, {1000 * dt / N:.3f} ms/frame")
