#!/bin/bash
# A/B runs of bench.py inside ONE gpurun call (boxes differ by several per cent: only numbers of one call compare).  Replaces round 2-5's family of ab_*.sh scripts.
#   tools/ab.sh [-w driver|long|full] [-r ROUNDS] [-b "SRC[ SRC..]:extra hipcc flags"] SETTING [SETTING ...] [-- bench.py arguments]
# A SETTING is "-" (defaults) or a comma-joined list of
#   name=value      an ifx_set_option switch                 (bench.py --opt name=value)
#   NAME=value      an environment variable (upper case)     e.g. IFX_OPTS=..., HIP_FORCE_DEV_KERNARG=0
#   lib:PATH        another build of libifx.so               (IFX_LIB=PATH; "lib:variant" = the library -b built)
#   tree:NAME       bench.py and library of another commit, exported and built under tools/ab/NAME
#                   (git archive <commit> bench.py instancefusion_amd include | tar -x -C tools/ab/NAME; make -C tools/ab/NAME/instancefusion_amd/csrc)
# -w: driver = the driver-shaped window (20 steps after 5, one segmentation call inside), long = 200 steps after 30 (default), full = the default run with every extra leg.
# -b: compile SRC.hip (e.g. ifx_track, or "ifx_track ifx_map") with the extra flags into scratch objects and link /tmp/libifx_variant.so; the in-tree library is not touched.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
ROOT=$PWD; window=long; rounds=2; build=""
while getopts "w:r:b:" o; do case $o in w) window=$OPTARG;; r) rounds=$OPTARG;; b) build=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
sets=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" = "--" ] && shift
case $window in
  driver) wargs="--steps 20 --warmup 5 --no-cpu-baseline --extras-frames 0";;
  long)   wargs="--steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0";;
  full)   wargs="--no-cpu-baseline";;
  *) echo "unknown window $window"; exit 2;;
esac
if [ -n "$build" ]; then
  SRC=${build%%:*}; X=${build#*:}
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -Wno-unused-value -Wno-unused-result"
  OBJS=""; for o in $(cd instancefusion_amd/csrc && ls *.hip | sed "s/\.hip$//"); do if [[ " $SRC " == *" $o "* ]]; then OBJS="$OBJS /tmp/v_$o.o"; else OBJS="$OBJS $o.o"; fi; done
  ( cd instancefusion_amd/csrc && for f in $SRC; do /opt/rocm/bin/hipcc $F $X -c $f.hip -o /tmp/v_$f.o || exit 1; done && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libifx_variant.so $OBJS -ldl ) || exit 1
fi
for r in $(seq $rounds); do
  for s in "${sets[@]}"; do
    opts=""; envs=""; dir=$ROOT
    if [ "$s" != "-" ]; then
      for item in ${s//,/ }; do
        case $item in
          lib:variant) envs="$envs IFX_LIB=/tmp/libifx_variant.so";;
          lib:*) envs="$envs IFX_LIB=$(realpath ${item#lib:})";;
          tree:*) dir=$ROOT/tools/ab/${item#tree:};;
          [A-Z]*=*) envs="$envs $item";;
          *) opts="$opts --opt $item";;
        esac
      done
    fi
    ( cd $dir && env $envs python bench.py $wargs $opts "$@" 2>/dev/null ) | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels']
us = {x: round(k[x]['avg_ms'] * 1000, 1) for x in ('icp_residual', 'rgb_step_solve', 'clean_raster_view', 'index_list', 'splat_resolve', 'index_resolve', 'append_scan', 'associate', 'fuse_update', 'bilateral_metric') if x in k}
lv = {e['kernel']: round(e['avg_launch_ms'] * 1000, 1) for e in d['roofline'].get('tracker_levels', [])}
g = d['ms_per_frame_gpu']
out = [sys.argv[1], d['value'], 'track', g['track'], 'fuse', g['fuse'], 'call', d['instance']['ms_per_call']]
if 'value_host_entry' in d:
    out += ['host', d['value_host_entry']['value'], d['value_host_entry_async']['value'], d['value_host_entry_async']['frames_only'], d.get('value_host_entry_hinted', {}).get('value'),
            'fast', d['value_fast_cadence']['value'], 'lc', d['value_close_loops']['value'], 'sharded', d['value_sharded']['value']]
print(*out, us, lv)" "$s"
  done
done
