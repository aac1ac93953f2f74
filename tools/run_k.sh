cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "loop_closure or closure or fern or deform or lookahead" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_y_tests.log | head -5
for r in 1 2; do for a in 0 1; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 --close-loops --opt lc_view=$a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lc_view=$a', d['value'], d['ms_per_frame_gpu'])"; done; done
