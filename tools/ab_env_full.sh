#!/bin/bash
# environment settings against each other on the default bench run (driver-shaped window + the extra legs) in ONE gpurun call: tools/ab_env_full.sh "VAR=1 -" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
sets=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for s in $sets; do
    if [ "$s" = "-" ]; then e=""; else e="$s"; fi
    env $e python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$s', d['value'], d['ms_per_step'], 'host', d['value_host_entry']['value'], d['value_host_entry_async']['value'], d['value_host_entry_async']['frames_only'], d['value_host_entry_hinted']['value'], 'fast', d['value_fast_cadence']['value'], 'lc', d['value_close_loops']['value'], 'sharded', d['value_sharded']['value'], 'call', d['instance']['ms_per_call'])"
  done
done
