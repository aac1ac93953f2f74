cd $GRAFT_REPO_ROOT
python bench.py --steps 3000 --warmup 30 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3000 frames', d['value'], d['ms_per_frame_gpu'], d['instance'], d['view_list'], d['ate_rms_m'], d['config'].get('surfels_live'), d['config'].get('surfel_slots'))"
python bench.py --steps 600 --warmup 30 --no-cpu-baseline --extras-frames 0 --close-loops 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('600 frames closeLoops', d['value'], d['ms_per_frame_gpu'], d['ate_rms_m'])"
