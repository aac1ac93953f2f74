cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "flood or segmentation or instance or full_loop or end_to_end or owner or resolutions" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed" gpurun_out/r03_y_tests.log
for i in 1 2 3; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'], {x:round(k[x]['avg_ms']*1000,1) for x in ('project_bbox','count_colour_px','sp_connect','sp_edges','ff_local','ff_relax') if x in k})"; done
