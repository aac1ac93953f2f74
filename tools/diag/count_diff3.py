import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import instancefusion_amd as ifx
from instancefusion_amd import synth
from instancefusion_amd.sharded import _DevWords
W, H, NF = 640, 480, 40
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
st = synth.make_stream(NF, W, H, noise=True, loop_len=90, **K)
a = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, **K); a.set_option("compact_every_frame", 1); a.set_option("view_list", 0)
b = ifx.ElasticFusion(w=W, h=H, max_surfels=3_000_000, **K); b.set_option("view_list", 0)
def live(e):
    v = e.map_view(); n = v.count
    tm = torch.as_tensor(_DevWords(v.d_times, n * 2, "<f4"), device="cuda").view(n, 2).cpu().numpy()
    ic = torch.as_tensor(_DevWords(v.d_img_corr, n * 4, "<f4"), device="cuda").view(n, 4).cpu().numpy()
    pc = torch.as_tensor(_DevWords(v.d_pos_conf, n * 4, "<f4"), device="cuda").view(n, 4).cpu().numpy()
    al = tm[:, 1] > -1.0e9
    return n, al, ic, tm, pc
for i in range(NF):
    a.processFrame(st["rgb"][i], st["depth"][i]); b.processFrame(st["rgb"][i], st["depth"][i])
    na, nb, sb = a.count, b.count, b.slots
    n_b, al_b, ic_b, tm_b, pc_b = live(b)
    tomb = int((~al_b).sum())
    if na != nb or sb - nb != tomb:
        print("frame", i, "eager", na, "lazy count", nb, "slots", sb, "n_dead by state", sb - nb, "tombstones by scan", tomb)
        n_a, al_a, ic_a, tm_a, pc_a = live(a)
        ka = {tuple(r) for r in ic_a[al_a][:, :3].astype(np.int64).tolist()}; kb = {tuple(r) for r in ic_b[al_b][:, :3].astype(np.int64).tolist()}
        print(" live sets equal:", ka == kb, len(ka), len(kb), "dups eager", int(al_a.sum()) - len(ka), "dups lazy", int(al_b.sum()) - len(kb))
        break
else:
    print("no mismatch in", NF, "frames")
