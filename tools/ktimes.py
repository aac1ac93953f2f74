"""Per-kernel HIP-event timing table of the bench workload (debug aid): python tools/ktimes.py [surfels] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(40, W, H, noise=True, loop_len=90, **K)
m = synth.make_map(n, st["scene"], st["poses_world"][0], 1000)
ef = ifx.ElasticFusion(w=W, h=H, max_surfels=n + 1_500_000, **K)
ef.processFrame(st["rgb"][0], st["depth"][0]); ef.upload(m); ef.set_pose(st["poses"][0], 1000); ef.combined_predict(st["poses"][0], 1000, 1000)
for i in range(1, 11): ef.processFrame(st["rgb"][i], st["depth"][i])
ef.set_option("kernel_timing", 1); ef.kernel_ms("__reset__")
for i in range(11, 11 + frames): ef.processFrame(st["rgb"][i % 40], st["depth"][i % 40])
ef.kernel_ms("__list__")
print("frames", frames, "slots", ef.slots)
import ctypes as C
L = ifx.lib()
if hasattr(L, "ifx_debug_stamps"):
    a = (C.c_longlong * 16)()
    L.ifx_debug_stamps(ef.handle, a, 0)
    v = list(a); n = max(v[2], 1)
    print("rgb body of the last block: sigma %.0f  main %.0f  block_reduce %.0f" % (v[8] / n, v[9] / n, v[10] / n))
    print("stamps per solve launch (cycles): rgb+handoff %.0f  solve_block %.0f  serial_part %.0f  rgb_body %.0f  partial_loads %.0f  barrier1 %.0f  lds_sum+barrier2 %.0f (n=%d)" % (v[0] / n, v[1] / n, v[4] / n, v[5] / n, v[3] / n, v[6] / n, v[7] / n, v[2]))
