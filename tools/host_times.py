#!/usr/bin/env python3
"""Host-side cost of the per-frame calls (debug aid): how long the CPU spends enqueuing a frame vs. how long the GPU needs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 90
st = synth.make_stream(L, W, H, noise=True, loop_len=90, **K)
m = synth.make_map(n, st["scene"], st["poses_world"][0], 1000)
ef = ifx.ElasticFusion(w=W, h=H, max_surfels=n + 1_500_000, **K)
inst = ifx.InstanceFusion(ef)
d_rgb = torch.from_numpy(st["rgb"]).cuda(); d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
ef.processFrame(st["rgb"][0], st["depth"][0]); ef.upload(m); ef.set_pose(st["poses"][0], 1000); ef.combined_predict(st["poses"][0], 1000, 1000)
pos = [1]
def run(frames, seg):
    t_enq = t_seg = 0.0
    t0 = time.perf_counter()
    k0 = pos[0]; pos[0] += frames
    for k in range(k0, k0 + frames):
        a = time.perf_counter()
        ef.hint_next_frame_device(d_rgb[(k + 1) % L].data_ptr(), d_dep[(k + 1) % L].data_ptr())
        ef.enqueue_frame_device(d_rgb[k % L].data_ptr(), d_dep[k % L].data_ptr(), k)
        b = time.perf_counter()
        if seg: inst.whetherDoSegmentation(100 + k)
        c = time.perf_counter()
        t_enq += b - a; t_seg += c - b
    ef.sync()
    t = time.perf_counter() - t0
    print(f"frames {frames} seg={seg}: {1e6*t/frames:7.1f} us/frame wall | host enqueue {1e6*t_enq/frames:6.1f} us | whetherDoSegmentation wait {1e6*t_seg/frames:6.1f} us")
run(30, True)
run(80, True)
run(80, False)
