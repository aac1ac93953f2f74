#!/bin/bash
# the round's full check on the GPU box: the GPU suite, [the one-off 640x480 sweep: SWEEP640=N], then the driver-shaped bench line -> gpurun_out/<tag>_gputests.txt, <tag>_bench_driver.json
TAG=${1:-r06_x}
cd ${GRAFT_REPO_ROOT:-.}
# (the full output goes to a file as it comes: a call that runs into its time limit still says how far it got; --durations names the slow tests)
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${TAG}_gputests_full.txt 2>&1
grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/${TAG}_gputests_full.txt | grep -v "^$" | tail -16 > gpurun_out/${TAG}_gputests.txt
cat gpurun_out/${TAG}_gputests.txt
if [ -n "$SWEEP640" ]; then
  IFX_SWEEP_EXTRA_640=$SWEEP640 python -m pytest tests/test_gpu_sweep.py -m gpu -q -k resident_frame_path_at_640x480 --durations=4 > gpurun_out/${TAG}_sweep640_full.txt 2>&1
  grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/${TAG}_sweep640_full.txt | grep -v "^$" | tail -8 > gpurun_out/${TAG}_sweep640.txt
  cat gpurun_out/${TAG}_sweep640.txt
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt   # (what the driver runs before the bench)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_per_frame_gpu"], "host", d["value_host_entry"]["value"], d["value_host_entry_async"]["value"], d["value_host_entry_async"]["frames_only"], d.get("value_host_entry_hinted", {}).get("value"), "fast", d["value_fast_cadence"]["value"], "lc", d["value_close_loops"]["value"], "sharded", d["value_sharded"]["value"], "guard", d["exact_sum_range_exceeded"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("parity_in_bench"))
PY
