cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "gn_prologue or persistent_level or tracker_and_map_configurations or end_to_end or loop_closure_detection or lookahead or fern_hooks or eviction" > gpurun_out/r04_d_tests.log 2>&1; tail -4 gpurun_out/r04_d_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_d_bench_driver.json 2>/dev/null
python bench.py --no-cpu-baseline --extras-frames 0 > gpurun_out/r04_d_bench.json 2>/dev/null
python - <<PY
import json
for f in ("r04_d_bench.json","r04_d_bench_driver.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], r["kernel"], r["frac"], {k:v["avg_ms"] for k,v in r["kernels"].items() if k in ("icp_residual","rgb_step_solve","gn_level")})
PY
