cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r03_zz_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_zz_tests.log | head -5
python bench.py --sharded --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r03_zz_bench_sharded_world_of_one.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r03_zz_bench_sharded_world_of_one.json").read().strip().splitlines()[-1])
print("sharded world of one", d["value"], d.get("exchange"), d["ms_per_frame_gpu"])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
