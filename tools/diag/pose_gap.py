#!/usr/bin/env python3
"""Where does the HIP / oracle pose gap come from?  One tracked frame from identical 5M-surfel maps, in several tracker configurations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import instancefusion_amd as ifx
import oracle_lib as ol
from instancefusion_amd import synth

W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
st = synth.make_stream(3, W, H, noise=True, loop_len=90, **K)
big = synth.make_map(N, st["scene"], st["poses_world"][0], 1000)
ol.set_threads(ol.usable_cores())
for name, kw in (("default", {}), ("no_so3", dict(so3=0)), ("icp_only", dict(icp_weight=100.0)), ("icp_only_no_so3", dict(icp_weight=100.0, so3=0)), ("no_pyramid", dict(pyramid=0))):
    g = ifx.ElasticFusion(w=W, h=H, max_surfels=N + 600_000, **K, **kw)
    o = ol.Oracle(w=W, h=H, max_surfels=N + 600_000, **K, **kw)
    g.processFrame(st["rgb"][0], st["depth"][0]); o.process_frame(st["rgb"][0], st["depth"][0])
    g.upload(big); o.upload(big)
    g.set_pose(st["poses"][0], 1000); o.set_pose(st["poses"][0], 1000)
    g.combined_predict(st["poses"][0], 1000, 1000); o.combined_predict(st["poses"][0], 1000, 1000)
    pg = g.processFrame(st["rgb"][1], st["depth"][1]); po = o.process_frame(st["rgb"][1], st["depth"][1])
    dg = g.tracker_diag()
    od = o.tracker_diag()
    gt = st["poses"][1]
    print(f"{name:18s} |dt(hip-orc)| {np.linalg.norm(pg[:3,3]-po[:3,3]):.2e}  |dt(hip-gt)| {np.linalg.norm(pg[:3,3]-gt[:3,3]):.2e}  |dt(orc-gt)| {np.linalg.norm(po[:3,3]-gt[:3,3]):.2e}  dR {np.abs(pg[:3,:3]-po[:3,:3]).max():.2e}  \n      hip diag {dg}\n      orc diag {od[:6]}")
    g.close(); o.close()
