#!/usr/bin/env python3
"""Per-kernel HIP-event times of the superpixel stages (640x480 synthetic frame)."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import instancefusion_amd as ifx  # noqa: E402
from instancefusion_amd import synth  # noqa: E402

K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(2, 640, 480, noise=True, **K)
g = ifx.ElasticFusion(w=640, h=480, max_surfels=1000, **K)
inst = ifx.InstanceFusion(g)
rgb, dep = st["rgb"][1], st["depth"][1]
masks, cls = synth.canned_masks(st["obj"][1], st["scene"])
for rep in range(3):
    seg, n = inst.gSLICrInterface(rgb)
    s2, fin, info = inst.mergeSuperPixel(dep, seg)
    out = inst.maskSuperPixelFilter_OverSeg(fin, masks)
t = time.perf_counter()
for rep in range(10):
    seg, n = inst.gSLICrInterface(rgb)
t1 = time.perf_counter()
for rep in range(10):
    s2, fin, info = inst.mergeSuperPixel(dep, seg)
t2 = time.perf_counter()
for rep in range(10):
    out = inst.maskSuperPixelFilter_OverSeg(fin, masks)
t3 = time.perf_counter()
print(f"wall ms per call (host buffers in/out): slic {100*(t1-t):.3f}  merge {100*(t2-t1):.3f}  filter({masks.shape[0]} masks) {100*(t3-t2):.3f}")
g.set_option("kernel_timing", 1)
g.kernel_ms("__reset__")
for rep in range(10):
    seg, n = inst.gSLICrInterface(rgb)
    s2, fin, info = inst.mergeSuperPixel(dep, seg)
    out = inst.maskSuperPixelFilter_OverSeg(fin, masks)
g.sync()
tot = 0.0
for nme in ["slic_cvt", "slic_init", "slic_assoc", "slic_update", "slic_enforce", "sp_gauss", "sp_posnor", "sp_sums", "sp_first_avg", "sp_recluster", "sp_second_avg", "sp_final",
            "sp_count", "sp_filter"]:
    avg, cnt = g.kernel_ms(nme)
    tot += avg * cnt / 10
    print(f"{nme:16s} avg {avg*1000:8.2f} us  x{cnt/10:.0f} per call")
print(f"kernels per refinement: {tot*1000:.1f} us")
