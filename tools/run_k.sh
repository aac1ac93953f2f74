cd $GRAFT_REPO_ROOT
T=r04_final
( time python -m pytest tests -m gpu -q ) > gpurun_out/${T}_tests.log 2>&1; tail -4 gpurun_out/${T}_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/pmc_collect.sh ${T} 2>&1 | tail -3
cp gpurun_out/${T}_pmc_traffic.json profiles/
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2>gpurun_out/${T}_bench.err
python bench.py > gpurun_out/${T}_bench.json 2>>gpurun_out/${T}_bench.err
python - <<PY
import json
for f in ("bench_driver","bench"):
    d=json.loads(open("gpurun_out/${T}_%s.json" % f).read().strip().splitlines()[-1])
    fc=d.get("value_fast_cadence") or {}
    vs=d.get("value_sharded") or {}
    print(f, d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], fc.get("value"), fc.get("instance_ms_per_call"), vs.get("value"), (vs.get("config5_world_of_one") or {}).get("ahead"), d.get("value_host_entry",{}).get("value"), d.get("value_close_loops",{}).get("value"), d["cpu_baseline"].get("parity_in_bench"), d["roofline"]["traffic_source"], d["roofline"]["frac"])
PY
bash tools/prof_run.sh ${T} > /dev/null 2>&1; head -3 gpurun_out/${T}_seg_call_timeline.txt; tail -2 gpurun_out/${T}_seg_call_timeline.txt
