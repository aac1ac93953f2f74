import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
W, H, NF = 640, 480, 90
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(NF, W, H, noise=True, loop_len=NF, **K)
a = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, **K); a.set_option("compact_every_frame", 1); a.set_option("view_list", 0)
b = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, **K); b.set_option("view_list", 0)
div = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b.set_option("compact_divisor", div)
for i in range(NF):
    a.processFrame(st["rgb"][i], st["depth"][i]); b.processFrame(st["rgb"][i], st["depth"][i])
    na, nb, sb = a.count, b.count, b.slots
    if na != nb or i % 10 == 0: print(i, na, nb, sb)
    if na != nb:
        ma, mb = a.download(), b.download()
        ka = {tuple(r): j for j, r in enumerate(ma["ic"][:, :3].astype(np.int64).tolist())}; kb = {tuple(r): j for j, r in enumerate(mb["ic"][:, :3].astype(np.int64).tolist())}
        print(" only eager:", [(k, ma["pc"][ka[k]][3], ma["tm"][ka[k]].tolist()) for k in sorted(set(ka) - set(kb))[:6]])
        print(" only lazy :", [(k, mb["pc"][kb[k]][3], mb["tm"][kb[k]].tolist()) for k in sorted(set(kb) - set(ka))[:6]])
        common = [k for k in ka if k in kb]
        ia = np.array([ka[k] for k in common]); ib = np.array([kb[k] for k in common])
        for f in ("pc", "nr", "tm", "col"):
            d = np.nonzero((ma[f][ia] != mb[f][ib]).any(axis=1))[0]
            print(" field", f, "differs on", len(d), "common surfels", [(common[j], ma[f][ia[j]].tolist(), mb[f][ib[j]].tolist()) for j in d[:3]])
        break
