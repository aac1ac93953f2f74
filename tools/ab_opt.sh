#!/bin/bash
# A/B of one ifx_set_option switch inside ONE gpurun call (boxes differ by several per cent): tools/ab_opt.sh name [rounds] [bench args...]
# prints frames/s and the stage times with name=0 and name=1 alternating.
cd ${GRAFT_REPO_ROOT:-.}
name=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do
  for v in 0 1; do
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 --opt $name=$v "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$name=$v', d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'], {x:round(k[x]['avg_ms']*1000,1) for x in ('icp_residual','rgb_step_solve','splat_resolve','raster_view','clean_view','clean_raster_view','index_list','index_resolve','append_scan','associate','fuse_update') if x in k})"
  done
done
