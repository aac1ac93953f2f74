cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -q ) > gpurun_out/r04_head_tests.log 2>&1; tail -4 gpurun_out/r04_head_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_head_bench_driver.json 2>gpurun_out/r04_head_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_head_bench_driver.json").read().strip().splitlines()[-1])
fc=d.get("value_fast_cadence") or {}; vs=d.get("value_sharded") or {}
print("driver", d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], fc.get("value"), vs.get("value"), (vs.get("config5_world_of_one") or {}).get("ahead"), d.get("value_host_entry",{}).get("value"), d["cpu_baseline"].get("parity_in_bench",{}).get("not_bit_equal"))
PY
