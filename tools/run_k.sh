cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "flood or segmentation or instance or end_to_end or config3 or config4 or full_loop or many_masks or superpixel or other_resolutions" > gpurun_out/r03_o_tests.log 2>&1; tail -4 gpurun_out/r03_o_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_o_bench_driver.json 2>/dev/null
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 > gpurun_out/r03_o_bench.json 2>/dev/null
python - <<PY
import json
for n in ("_driver",""):
    f="gpurun_out/r03_o_bench%s.json" % n
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    print(n or "200", d["value"], d["ms_per_frame_gpu"], d["instance"]["ms_per_call"], d.get("value_host_entry",{}).get("value"), d.get("value_close_loops",{}).get("value"))
PY
tools/prof_run.sh r03_o > /dev/null 2>&1; grep -v "k_vote_update_m\|fillBuffer\|copyBuffer" gpurun_out/r03_o_seg_call_timeline.txt | head -70
