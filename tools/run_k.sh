cd $GRAFT_REPO_ROOT
T=r04_zz_final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2>gpurun_out/${T}_bench.err
python bench.py > gpurun_out/${T}_bench.json 2>>gpurun_out/${T}_bench.err
python - <<PY
import json
for f in ("bench_driver","bench"):
    d=json.loads(open("gpurun_out/${T}_%s.json" % f).read().strip().splitlines()[-1])
    fc=d.get("value_fast_cadence") or {}
    vs=d.get("value_sharded") or {}
    print(f, d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], fc.get("value"), vs.get("value"), (vs.get("config5_world_of_one") or {}).get("in_frame"), (vs.get("config5_world_of_one") or {}).get("ahead"), d["cpu_baseline"].get("parity_in_bench",{}).get("not_bit_equal"), d["roofline"]["traffic_source"])
PY
