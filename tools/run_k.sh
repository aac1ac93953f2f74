cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "not config4" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed\|skipped\|Error" gpurun_out/r03_y_tests.log | head
for a in "" "--seg-host-frame" "" "--seg-host-frame"; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'])"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-shaped', d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'])"
IFX_SEG_TRACE=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 2>&1 >/dev/null | grep "seg call" | tail -2
