cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "loop_closure or segmentation or lookahead or fern or deform" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_y_tests.log | head -5
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['instance']['ms_per_call'], d['value_host_entry']['value'], d['value_close_loops']['value'])"; done
python bench.py --no-cpu-baseline --close-loops --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('close-loops main leg', d['value'], d['ms_per_frame_gpu'])"
