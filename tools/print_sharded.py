"""one line per bench.py run: the sharded-map leg (stdin: the JSON line of bench.py; argv[1]: a tag)"""
import json
import sys

d = json.loads(sys.stdin.readline())
v = d["value_sharded"]
x = v["exchange"]
print(sys.argv[1] if len(sys.argv) > 1 else "-", "value", d["value"], "sharded", v["value"], v["ms_per_frame_gpu"], "collectives/frame", x["collectives_per_frame"], "B/px/frame", x["bytes_per_pixel_per_frame"],
      "calls", v["segmentation_calls"], "tracker_rows", v.get("tracker_rows", {}).get("value"), v.get("tracker_rows", {}).get("collectives_per_frame"), "config5(1 GPU)", v.get("config5_world_of_one", {}).get("value"))
