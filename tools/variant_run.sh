#!/bin/bash
# bench of a compile-time variant of one kernel file (run on the GPU box): SRC=ifx_map tools/variant_run.sh "<extra hipcc flags>" [bench options...]
# rebuilds $SRC.o (default ifx_track) with the extra flags into a scratch object, relinks the library, runs the bench, restores the regular library
cd "$(dirname "$0")/.."
X="$1"; shift
SRC=${SRC:-ifx_track}     # one file, or several: SRC="ifx_track ifx_map ifx_api"
OBJS=""; for o in ifx_api ifx_track ifx_map ifx_instance ifx_slic ifx_knn; do if [[ " $SRC " == *" $o "* ]]; then OBJS="$OBJS /tmp/v_$o.o"; else OBJS="$OBJS $o.o"; fi; done
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-unused-result"
cp instancefusion_amd/libifx.so /tmp/libifx.keep
( cd instancefusion_amd/csrc && for f in $SRC; do /opt/rocm/bin/hipcc $F $X -c $f.hip -o /tmp/v_$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libifx.so $OBJS ) || { cp /tmp/libifx.keep instancefusion_amd/libifx.so; exit 1; }
python bench.py --gpus 1 --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('variant [$X] [$*]', d['value'], 'track', d['ms_per_frame_gpu']['track'], 'fuse', d['ms_per_frame_gpu']['fuse'], 'icp', k['icp_residual']['avg_ms'], 'rgb', k['rgb_step_solve']['avg_ms'], 'index', k['index_list']['avg_ms'], 'clean', k['clean_view']['avg_ms'], 'raster', k['raster_view']['avg_ms'])"
cp /tmp/libifx.keep instancefusion_amd/libifx.so
