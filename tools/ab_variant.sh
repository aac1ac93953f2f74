#!/bin/bash
# A/B of a compile-time variant against the built library inside ONE gpurun call: tools/ab_variant.sh <src (e.g. ifx_track)> "<extra hipcc flags>" [rounds] [bench args...]
cd ${GRAFT_REPO_ROOT:-.}
SRC=$1; X="$2"; rounds=${3:-2}; shift; shift; shift
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-unused-result"
OBJS=""; for o in $(cd instancefusion_amd/csrc && ls *.hip | sed "s/\.hip$//"); do if [[ " $SRC " == *" $o "* ]]; then OBJS="$OBJS /tmp/v_$o.o"; else OBJS="$OBJS $o.o"; fi; done
( cd instancefusion_amd/csrc && for f in $SRC; do /opt/rocm/bin/hipcc $F $X -c $f.hip -o /tmp/v_$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libifx_variant.so $OBJS -ldl ) || exit 1
for r in $(seq $rounds); do
  for v in base variant; do
    if [ $v = variant ]; then export IFX_LIB=/tmp/libifx_variant.so; else unset IFX_LIB; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$v [$X]', d['value'], d['ms_per_frame_gpu'], {x:round(k[x]['avg_ms']*1000,1) for x in ('icp_residual','rgb_step_solve','clean_raster_view','index_list','splat_resolve') if x in k})"
  done
done
