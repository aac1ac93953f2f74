#!/bin/bash
# option settings against each other on the DRIVER-shaped window (20 steps after 5 warm-up, one segmentation call inside) in ONE gpurun call:
#   tools/ab_driver.sh "a=1,b=2 -" [rounds]
cd ${GRAFT_REPO_ROOT:-.}
sets=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for s in $sets; do
    if [ "$s" = "-" ]; then o=""; else o=$(echo $s | sed 's/,/ --opt /g; s/^/--opt /'); fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extras-frames 0 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$s', d['value'], d['ms_per_step'], d['ms_per_frame_gpu'], 'call', d['instance']['ms_per_call'])"
  done
done
