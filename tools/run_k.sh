cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "segmentation or instance or full_loop" 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'])"; done
IFX_SEG_TRACE=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 2>&1 >/dev/null | grep "seg call" | tail -3
