"""Persistent Gauss-Newton level kernel against the two-launch form on the bench stream (run on the GPU box): poses bit for bit, barrier time-outs, frame rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import instancefusion_amd as ifx
from instancefusion_amd import synth
W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
NF = 24
st = synth.make_stream(NF, W, H, noise=True, loop_len=90, **K)
m = synth.make_map(300_000, st["scene"], st["poses_world"][0], 1000)
def run(persist):
    ef = ifx.ElasticFusion(w=W, h=H, max_surfels=1_500_000, **K)
    ef.set_option("gn_persist", persist)
    ef.processFrame(st["rgb"][0], st["depth"][0]); ef.upload(m); ef.set_pose(st["poses"][0], 1000); ef.combined_predict(st["poses"][0], 1000, 1000)
    poses = []
    t0 = time.perf_counter()
    for i in range(1, NF):
        poses.append(ef.processFrame(st["rgb"][i], st["depth"][i]).copy())
    ef.sync()
    dt = time.perf_counter() - t0
    ef.close()
    return np.stack(poses), dt
a, ta = run(0)
b, tb = run(1)
print("two-launch %.1f ms, persistent %.1f ms for %d frames (host-entry path)" % (ta * 1e3, tb * 1e3, NF - 1))
print("poses identical:", np.array_equal(a, b), " max |diff|", float(np.abs(a - b).max()))
