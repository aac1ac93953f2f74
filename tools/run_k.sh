cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r03_p_tests.log 2>&1; tail -6 gpurun_out/r03_p_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_p_bench_driver.json 2>/dev/null
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 > gpurun_out/r03_p_bench.json 2>/dev/null
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 --opt lazy_ids=0 > gpurun_out/r03_p_bench_eager.json 2>/dev/null
python - <<PY
import json
for n in ("_driver","","_eager"):
    f="gpurun_out/r03_p_bench%s.json" % n
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    print(n or "200", d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], {x:k[x]["avg_ms"] for x in ("raster_view","splat_resolve","raster_finish") if x in k}, d.get("value_host_entry",{}).get("value"), d.get("value_close_loops",{}).get("value"))
PY
