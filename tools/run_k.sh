cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q ) > gpurun_out/r04_j_tests.log 2>&1; tail -6 gpurun_out/r04_j_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_j_bench_driver.json 2>gpurun_out/r04_j_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_j_bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_frame_gpu"], d["instance"], d.get("value_fast_cadence"), d.get("value_sharded", {}).get("value"), d["cpu_baseline"].get("parity_in_bench"))
PY
bash tools/prof_run.sh r04_j > /dev/null 2>&1; head -3 gpurun_out/r04_j_seg_call_timeline.txt; tail -2 gpurun_out/r04_j_seg_call_timeline.txt
