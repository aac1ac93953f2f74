cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 --res 1280x960 --surfels 20000000 > gpurun_out/r03_r_bench_1280x960_20M.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r03_r_bench_1280x960_20M.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("1280x960 20M", d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], {x:k[x]["avg_ms"] for x in ("raster_view","clean_view","index_list","cull_frame") if x in k})
PY
for n in 1000000 2000000 5000000 10000000 20000000 50000000; do python bench.py --steps 150 --warmup 30 --no-cpu-baseline --extras-frames 0 --surfels $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($n, d['value'], d['ms_per_frame_gpu']['track'], d['ms_per_frame_gpu']['fuse'], d['instance']['ms_per_call'])"; done > gpurun_out/r03_r_size_sweep.txt 2>&1
cat gpurun_out/r03_r_size_sweep.txt
python tools/replay_bench.py --frames 480 > gpurun_out/r03_r_replay_bench.txt 2>&1; tail -6 gpurun_out/r03_r_replay_bench.txt
