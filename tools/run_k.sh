cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r03_final_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_final_tests.log | head -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/r03_final_bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final_bench_driver.json 2>/dev/null
python - <<PY
import json
for f in ("r03_final_bench.json","r03_final_bench_driver.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], r["kernel"], r["frac"], r.get("traffic"), d["value_host_entry"]["value"], d["value_close_loops"]["value"], d["cpu_baseline"]["value"])
PY
python tools/replay_bench.py --frames 480 > gpurun_out/r03_final_replay_bench.txt 2>&1; grep -o "^[^>]*-> 480 frames in [0-9.]* s ([0-9.]* frames/s" gpurun_out/r03_final_replay_bench.txt
