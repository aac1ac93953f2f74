#!/bin/bash
# bench of a compile-time variant of one kernel file (run on the GPU box): SRC=ifx_map tools/variant_run.sh "<extra hipcc flags>" [bench options...]
# builds $SRC.o (default ifx_track) with the extra flags into scratch objects, links a SCRATCH library (/tmp/libifx_variant.so, selected through IFX_LIB)
# and runs the bench on it; instancefusion_amd/libifx.so is never touched
cd "$(dirname "$0")/.."
X="$1"; shift
SRC=${SRC:-ifx_track}     # one file, or several: SRC="ifx_track ifx_map ifx_api"
OBJS=""; for o in $(cd instancefusion_amd/csrc && ls *.hip | sed "s/\.hip$//"); do if [[ " $SRC " == *" $o "* ]]; then OBJS="$OBJS /tmp/v_$o.o"; else OBJS="$OBJS $o.o"; fi; done
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-unused-result"
( cd instancefusion_amd/csrc && for f in $SRC; do /opt/rocm/bin/hipcc $F $X -c $f.hip -o /tmp/v_$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libifx_variant.so $OBJS ) || exit 1
IFX_LIB=/tmp/libifx_variant.so python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
g=lambda n: k.get(n,{}).get('avg_ms')
print('variant [$X] [$*]', d['value'], 'track', d['ms_per_frame_gpu']['track'], 'fuse', d['ms_per_frame_gpu']['fuse'], 'icp', g('icp_residual'), 'rgb', g('rgb_step_solve'), 'index', g('index_list'), 'clean', g('clean_view'), 'raster', g('raster_view'))"
