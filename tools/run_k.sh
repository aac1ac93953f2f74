cd $GRAFT_REPO_ROOT
for q in 4 8 4 8; do GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 200 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', d['value'], d['instance']['ms_per_call'], d['value_host_entry']['value'], d['value_close_loops']['value'])"; done
