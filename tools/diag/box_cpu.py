import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from instancefusion_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup v2 cpu.max", e)
W, H = 640, 480
K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
t = time.time(); st = synth.make_stream(6, W, H, noise=True, loop_len=90, **K); print("stream gen s/frame", (time.time() - t) / 6)
for th in (1, 4, 8, 16, 32, 64):
    ol.set_threads(th)
    o = ol.Oracle(w=W, h=H, max_surfels=1000000, **K)
    t = time.time()
    for i in range(6): o.process_frame(st["rgb"][i], st["depth"][i])
    print(th, "threads", (time.time() - t) / 6, "s/frame", o.stage_ms(reset=True))
    o.close()
