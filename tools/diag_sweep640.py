"""A case of tests/test_gpu_sweep.py::test_resident_frame_path_at_640x480_other_scenes frame by frame: live surfels of the HIP path (several option sets) against the oracle.
Usage: python tools/diag_sweep640.py SEED MOTION [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import instancefusion_amd as ifx
import oracle_lib as orc
from instancefusion_amd import synth

seed, motion = int(sys.argv[1]), sys.argv[2]
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 24
STOP_AT_FIRST = len(sys.argv) > 4
W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
CONF = 3.0
orc.build()
scene = synth.Scene(seed)
st = synth.make_stream_from_poses(synth.trajectory_profile(motion, NF, seed), scene, W, H, noise_seed=seed + 1, **K)
orc.set_threads(orc.usable_cores())
d_rgb = torch.from_numpy(st["rgb"]).cuda()
d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
torch.cuda.synchronize()
o = orc.Oracle(w=W, h=H, max_surfels=2_000_000, confidence=CONF, **K)
sets = {"default": {}, "clean_raster=0": {"clean_raster": 0}, "hot_records=0": {"hot_records": 0}, "view_list=0": {"view_list": 0}, "compact_every_frame": {"compact_every_frame": 1},
        "no hint": {"_nohint": 1}}
gs = {}
for name, opts in sets.items():
    g = ifx.ElasticFusion(w=W, h=H, max_surfels=2_000_000, confidence=CONF, **K)
    for k, v in opts.items():
        if not k.startswith("_"):
            g.set_option(k, v)
    gs[name] = (g, ifx.InstanceFusion(g), opts)
first_bad = {n: None for n in sets}
for i in range(NF):
    po = o.process_frame(st["rgb"][i], st["depth"][i])
    row = [f"frame {i:2d} oracle {o.count:7d}"]
    for name, (g, inst, opts) in gs.items():
        if i + 1 < NF and "_nohint" not in opts:
            g.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
        g.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
        inst.whetherDoSegmentation(100 + i)
        pg = g.trajectory(1)[0]
        c = g.count
        eq = np.array_equal(np.asarray(pg, np.float32), np.asarray(po, np.float32), equal_nan=True)
        row.append(f"{name}: {c - o.count:+d}{'' if eq else ' POSE'}")
        if c != o.count and first_bad[name] is None:
            first_bad[name] = i
    print(" | ".join(row), flush=True)
    if STOP_AT_FIRST and first_bad["default"] is not None:
        g = gs["default"][0]
        mg, mo_ = g.download(), o.download()
        a, b = mg["pc"], mo_["pc"]
        ia = ib = 0
        extra = []
        while ia < len(a) and ib < len(b):
            if np.array_equal(a[ia], b[ib]) and np.array_equal(mg["tm"][ia], mo_["tm"][ib]):
                ia += 1; ib += 1
            else:
                extra.append(ia); ia += 1
        extra += list(range(ia, len(a)))
        print("rows of the HIP map that the oracle's map does not hold:", len(extra), "of", len(a), "| oracle", len(b))
        Tinv = np.linalg.inv(np.asarray(po, np.float64))
        for r in extra[:40]:
            p = Tinv @ np.append(a[r, :3].astype(np.float64), 1.0)
            u, v = K["fx"] * p[0] / p[2] + K["cx"], K["fy"] * p[1] / p[2] + K["cy"]
            print(f"  row {r}: conf {a[r, 3]:.3f} tm {mg['tm'][r]} radius {mg['nr'][r, 3]:.4f} cam z {p[2]:.3f} pixel ({u:.1f}, {v:.1f})")
        tms = mg["tm"][extra]
        print("last-update times of the extra rows:", np.unique(tms[:, 1], return_counts=True), "init", np.unique(tms[:, 0], return_counts=True), "tick", o.tick)
        break
print("first frame with another count:", first_bad)
vs = gs["default"][0].view_list_stats()
print("view list of the default handle:", vs)
mo = o.download()
for name, (g, inst, opts) in gs.items():
    mg = g.download()
    if mg["pc"].shape == mo["pc"].shape:
        print(name, {k: bool(np.array_equal(mg[k], mo[k], equal_nan=True)) for k in mg})
    else:
        print(name, "shapes", mg["pc"].shape, mo["pc"].shape)
