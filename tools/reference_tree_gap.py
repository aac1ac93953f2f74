#!/usr/bin/env python3
"""How far is the reference's own arithmetic from the exact sums the parity claims rest on?  (VERDICT round 2, weak point 2.)

The oracle (CPU restatement) runs the 640x480 benchmark loop twice: with the exact, order-independent sums of the normal equations (what the HIP
path reproduces bit for bit) and with `orc_set_sum_order(3)`: f32 products summed in f32 in the tree the reference's kernels build with its GTX 1080
launch table (EF/Cuda/reduce.cu:133-185, :397-402; EF/Utils/GPUConfig.h:123-126).  Everything else -- pixels, gates, solver, map -- is identical.
Writes the per-frame trajectory gap as JSON (committed: profiles/archive/r03_reference_tree_gap.json).

    python tools/reference_tree_gap.py [frames=90] [out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(frames=90, W=640, H=480):
    import oracle_lib as ol
    from instancefusion_amd import synth

    ol.build()
    K = dict(fx=528.0 * W / 640, fy=528.0 * W / 640, cx=W / 2.0, cy=H / 2.0)
    st = synth.make_stream(frames, W, H, noise=True, loop_len=90, **K)
    ol.set_threads(ol.usable_cores())
    L = ol.lib()
    traj = {}
    for mode in (0, 3):
        L.orc_set_sum_order(mode)
        o = ol.Oracle(w=W, h=H, max_surfels=3_000_000, **K)
        traj[mode] = np.stack([o.process_frame(st["rgb"][i], st["depth"][i]).copy() for i in range(frames)])
        o.close()
    L.orc_set_sum_order(0)
    ol.set_threads(1)
    gap = np.linalg.norm(traj[0][:, :3, 3] - traj[3][:, :3, 3], axis=1)
    rot = np.abs(traj[0][:, :3, :3] - traj[3][:, :3, :3]).reshape(frames, -1).max(axis=1)
    gt = st["poses"][:frames]
    ate = {m: float(np.sqrt(np.mean(np.sum((traj[m][:, :3, 3] - gt[:, :3, 3]) ** 2, axis=1)))) for m in (0, 3)}
    return dict(frames=frames, resolution=f"{W}x{H}",
                what="oracle with exact order-independent sums (= the HIP path, bit for bit) against the oracle with the reference's f32 summation tree (GTX 1080 launch table); translation gap per frame in metres",
                gap_m=[float(g) for g in gap], rot_entry_gap=[float(r) for r in rot],
                gap_rms_m=float(np.sqrt(np.mean(gap ** 2))), gap_max_m=float(gap.max()), first_frame_over_1e_4=int(np.argmax(gap > 1e-4)) if (gap > 1e-4).any() else None,
                ate_vs_ground_truth_rms_m={"exact_sums": ate[0], "reference_tree": ate[3]})


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 90
    out = run(n)
    js = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(js + "\n")
    print({k: out[k] for k in ("gap_rms_m", "gap_max_m", "first_frame_over_1e_4", "ate_vs_ground_truth_rms_m")})
    print(["%.1e" % g for g in out["gap_m"]])
