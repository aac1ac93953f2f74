"""Quick GPU-side comparison of the HIP path against the CPU oracle on a short synthetic stream."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import instancefusion_amd as ifx
from instancefusion_amd import synth
import oracle_lib as ol

W, H = int(os.environ.get("W", 320)), int(os.environ.get("H", 240))
N = int(os.environ.get("N", 8))
s = W / 640.0
K = dict(fx=528.0 * s, fy=528.0 * s, cx=320.0 * s, cy=240.0 * s)
st = synth.make_stream(N, W, H, K["fx"], K["fy"], K["cx"], K["cy"], noise=True)
o = ol.Oracle(w=W, h=H, max_surfels=600000, **K)
g = ifx.ElasticFusion(w=W, h=H, max_surfels=600000, **K)
g.set_option("compact_every_frame", 1)
for i in range(N):
    t0 = time.time(); po = o.process_frame(st["rgb"][i], st["depth"][i]); t1 = time.time()
    pg = g.processFrame(st["rgb"][i], st["depth"][i]); t2 = time.time()
    df = np.abs(o.image("depth_filtered").astype(int) - g.image("depth_filtered").astype(int))
    ida, idb = o.image("ids_after"), g.image("ids_after")
    pv = np.abs(o.image("pred_vertex") - g.image("pred_vertex"))
    print(f"frame {i}: n_orc={o.count} n_gpu={g.count} |dpose|={np.abs(po-pg).max():.2e} depth_filt_maxdiff={df.max()} ndiff={(df>0).sum()} "
          f"ids_mismatch={(ida!=idb).mean():.4f} predv_maxdiff={np.nanmax(pv):.2e} cpu={t1-t0:.3f}s gpu={t2-t1:.3f}s diag={g.tracker_diag()[:4]}")
mo, mg = o.download(), g.download()
n = min(len(mo["pc"]), len(mg["pc"]))
print("map pc maxdiff over common prefix", np.abs(mo["pc"][:n] - mg["pc"][:n]).max())
print("stage ms", g.stage_ms())
