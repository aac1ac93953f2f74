cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "owner_sharded or config5_two or sharded_rccl or cpp_replay_sharded or config4" > gpurun_out/r04_g_tests.log 2>&1; tail -4 gpurun_out/r04_g_tests.log
python bench.py --sharded --no-cpu-baseline --extras-frames 0 > gpurun_out/r04_g_bench_sharded_world_of_one.json 2>gpurun_out/r04_g_bench_sh.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r04_g_bench_sharded_world_of_one.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_frame_gpu"], d.get("exchange"), d["view_list"])
PY
( time python -m pytest tests -m gpu -q -k "config5_8_streams" ) > gpurun_out/r04_g_config5.log 2>&1; tail -6 gpurun_out/r04_g_config5.log; free -g | head -2
