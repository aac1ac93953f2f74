#!/bin/bash
# the one-off WIDE parity sweeps of a round, on the GPU box (HIP path against the oracle well beyond the committed cases): tools/sweeps_one_off.sh <tag>   -> gpurun_out/<tag>_sweeps_one_off.txt
TAG=${1:-r06_x}
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/${TAG}_sweeps_one_off.txt
: > $OUT
run() {   # label, env assignments, pytest selection
  local label="$1"; shift; local envs="$1"; shift
  echo "== $label   ($envs python -m pytest $*)" >> $OUT
  env $envs python -m pytest "$@" -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|pose parity|^FAILED|^E  " | head -12 >> $OUT
}
run "8-frame sweep, 512 scene seeds x motion x degradation" "IFX_SWEEP_EXTRA=488" tests/test_gpu_sweep.py -k seed_sweep
run "resident-frame path at 640x480, 61 scenes x motions" "IFX_SWEEP_EXTRA_640=56" tests/test_gpu_sweep.py -k other_scenes
run "long runs (56 frames, a call every 9th), 12 scenes" "IFX_SWEEP_LONG=11" tests/test_gpu_sweep.py -k long_run
run "1280x960, 4 scenes" "IFX_SWEEP_EXTRA_1280=4" tests/test_gpu_sweep.py -k "other_scenes and 25"
run "sharded map on emulated ranks, 12 further scenes" "IFX_SWEEP_SHARDED=12" tests/test_gpu_parity.py -k owner_sharded_map_emulated
cat $OUT
