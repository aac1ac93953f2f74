#!/bin/bash
# several option settings against each other inside ONE gpurun call: tools/ab_opts.sh "a=1,b=2 a=3 -" [rounds]   ("-" = defaults; commas join options of one setting)
cd ${GRAFT_REPO_ROOT:-.}
sets=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for s in $sets; do
    if [ "$s" = "-" ]; then o=""; else o=$(echo $s | sed 's/,/ --opt /g; s/^/--opt /'); fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; lv={e['kernel']:round(e['avg_launch_ms']*1000,1) for e in d['roofline'].get('tracker_levels',[])}
print('$s', d['value'], d['ms_per_frame_gpu']['track'], d['ms_per_frame_gpu']['fuse'], lv)"
  done
done
