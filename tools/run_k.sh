cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "persistent or lookahead or full_loop or end_to_end or configurations or resolutions" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_y_tests.log | head -5
for i in 1 2 3; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'])"; done
