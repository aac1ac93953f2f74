cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "not config4" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed\|skipped" gpurun_out/r03_y_tests.log
for r in 1 2; do for a in 4 0; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 --opt gn_persist=$a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gn_persist=$a', d['value'], d['ms_per_frame_gpu'])"; done; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-shaped', d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'])"
