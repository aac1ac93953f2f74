"""Diagnostic: the 90-frame 640x480 loop under option variants; which one moves the final map away from the compact-every-frame run (== the oracle, test_full_loop)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
W, H, NF = 640, 480, 90
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(NF, W, H, noise=True, loop_len=NF, **K)
d_rgb = torch.from_numpy(st["rgb"]).cuda(); d_dep = torch.from_numpy(st["depth"].view(np.int16)).cuda()
def run(opts, hint, seg, should):
    g = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, **K)
    for k, v in opts.items(): g.set_option(k, v)
    inst = ifx.InstanceFusion(g)
    for i in range(NF):
        if hint and i + 1 < NF: g.hint_next_frame_device(d_rgb[i + 1].data_ptr(), d_dep[i + 1].data_ptr())
        g.enqueue_frame_device(d_rgb[i].data_ptr(), d_dep[i].data_ptr(), i)
        if should: inst.whetherDoSegmentation(100 + i)
        if seg and i in (35, 60, 89):
            masks, cls = synth.canned_masks(st["obj"][i], st["scene"])
            inst.ProcessSegmentation(None, None, masks, cls, i, superpixels=True)
    n = g.count; m = g.download(); g.close()
    return n, m
base_n, base = run(dict(compact_every_frame=1), False, False, False)
print("base", base_n)
for name, (o, hint, seg, should) in dict(lazy=(dict(), False, False, False), lazy_novl=(dict(view_list=0), False, False, False), lazy_hint=(dict(), True, False, False),
                                 lazy_should=(dict(), False, False, True), lazy_seg=(dict(), False, True, True), all=(dict(), True, True, True), all_fullids=(dict(lazy_ids=0), True, True, True)).items():
    n, m = run(o, hint, seg, should)
    same = n == base_n and all(np.array_equal(m[k], base[k]) for k in ("pc", "nr", "tm"))
    print(name, n, "same" if same else "DIFFERENT")
    if n != base_n:
        a = {tuple(r) for r in base["ic"][:, :3].astype(np.int64).tolist()}; b = {tuple(r) for r in m["ic"][:, :3].astype(np.int64).tolist()}
        miss = sorted(a - b)[:8]
        print("  missing (px, py, frame of creation):", miss)
        for (x, y, f) in miss[:4]:
            j = np.nonzero((base["ic"][:, 0].astype(np.int64) == x) & (base["ic"][:, 1].astype(np.int64) == y) & (base["ic"][:, 2].astype(np.int64) == f))[0][0]
            print("   base row", j, "pc", base["pc"][j], "tm", base["tm"][j])
