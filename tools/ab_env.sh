#!/bin/bash
# A/B of an environment variable inside ONE gpurun call: tools/ab_env.sh NAME "v1 v2 ..." [rounds] [bench args...]   ("-" = unset)
cd ${GRAFT_REPO_ROOT:-.}
name=$1; vals=$2; rounds=${3:-2}; shift; shift; shift
for r in $(seq $rounds); do
  for v in $vals; do
    if [ "$v" = "-" ]; then unset $name; else export $name=$v; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$name=$v', d['value'], d['ms_per_frame_gpu'], {x:round(k[x]['avg_ms']*1000,1) for x in ('icp_residual','rgb_step_solve','clean_raster_view','index_list','splat_resolve','bilateral_metric') if x in k})"
  done
done
