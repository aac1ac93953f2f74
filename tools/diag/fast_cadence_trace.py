"""Host times of the steps of bench.py's fast-cadence leg (a map started from nothing, a call every 3rd frame): per frame the time of enqueue_frame_device,
of whetherDoSegmentation and of the call, so that a slow leg can be pinned on the step that is slow.  Usage: python tools/diag/fast_cadence_trace.py [opt=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import instancefusion_amd as ifx  # noqa: E402
from instancefusion_amd import synth  # noqa: E402

W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
L = 90
st = synth.make_stream(L, W, H, noise=True, loop_len=L, seed=synth.SEED, **K)
masks = [synth.canned_masks(st["obj"][i], st["scene"]) for i in range(L)]
d_rgb = torch.from_numpy(st["rgb"]).cuda()
d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
torch.cuda.synchronize()
for rep in range(2):
    ef = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, **K)
    inst = ifx.InstanceFusion(ef)
    for kv in sys.argv[1:]:
        k_, v_ = kv.split("=")
        ef.set_option(k_, int(v_))
    rows = []
    t_all = None
    for i in range(L):
        if i == 12:
            ef.sync(); torch.cuda.synchronize()
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        if i + 1 < L:
            ef.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
        ef.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
        t1 = time.perf_counter()
        f = inst.whetherDoSegmentation(100 + i)
        t2 = time.perf_counter()
        if f and masks[i][0].shape[0]:
            inst.ProcessSegmentation(None, None, masks[i][0], masks[i][1], i, superpixels=True)
        t3 = time.perf_counter()
        rows.append((i, int(f), (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
    ef.sync(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all
    print(f"rep {rep}: {(L - 12) / t_all:.1f} frames/s; ahead {ef.superpixel_ahead_stats()}", file=sys.stderr)
    if rep == 1:
        for r in rows[12:60]:
            print("frame %2d fired %d  enqueue %6.0f  decide %6.0f  call %6.0f us" % r, file=sys.stderr)
    ef.close()
