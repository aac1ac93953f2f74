#!/usr/bin/env python3
"""Pretty-prints the kernel table of a bench.py JSON line read from stdin."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], d["ms_per_frame_gpu"])
ks = d["roofline"]["kernels"]
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
    print("%-18s %7.1f us x %4.1f = %7.1f us/frame" % (k, v["avg_ms"] * 1000, v["launches"] / 20, v["avg_ms"] * v["launches"] * 50))
for e in d["roofline"].get("streaming_passes", []):
    print(e["kernel"], e["achieved"], e["frac"], e["traffic"])
