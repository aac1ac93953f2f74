cd $GRAFT_REPO_ROOT
T=r04_y
( time python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2>gpurun_out/${T}_bench.err
python bench.py > gpurun_out/${T}_bench.json 2>>gpurun_out/${T}_bench.err
python - <<PY
import json
for f in ("bench_driver","bench"):
    d=json.loads(open("gpurun_out/${T}_%s.json" % f).read().strip().splitlines()[-1])
    fc=d.get("value_fast_cadence") or {}
    print(f, d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], fc.get("value"), fc.get("instance_ms_per_call"), d.get("value_sharded", {}).get("value"), d.get("value_host_entry",{}).get("value"), d.get("value_close_loops",{}).get("value"), d["cpu_baseline"].get("parity_in_bench"))
PY
bash tools/prof_run.sh ${T} > /dev/null 2>&1; head -3 gpurun_out/${T}_seg_call_timeline.txt; tail -3 gpurun_out/${T}_seg_call_timeline.txt
