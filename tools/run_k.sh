cd $GRAFT_REPO_ROOT
tools/pmc_collect.sh r03_q > gpurun_out/r03_q_pmc.log 2>&1
tools/pmc_collect.sh r03_q_morton --map-order morton > gpurun_out/r03_q_morton_pmc.log 2>&1
tools/prof_run.sh r03_q > gpurun_out/r03_q_prof.log 2>&1
python bench.py --steps 200 --warmup 30 --extras-frames 0 --no-cpu-baseline --map-order morton > gpurun_out/r03_q_bench_morton.json 2>/dev/null
python bench.py --steps 200 --warmup 30 > gpurun_out/r03_q_bench.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_q_bench_driver.json 2>/dev/null
tail -3 gpurun_out/r03_q_pmc.log
python - <<PY
import json
for n in ("","_morton","_driver"):
    d=json.loads(open("gpurun_out/r03_q_bench%s.json" % n).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(n or "200", d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], r["kernel"], r["frac"], r["traffic"], [(e["kernel"], e["frac"], e["traffic_over_algorithmic"]) for e in r["map_passes"][:3]], r["stages"])
PY
