#!/bin/bash
# the tree of another commit (exported and built under tools/ab/<name>: git archive <commit> bench.py instancefusion_amd include | tar -x -C tools/ab/<name>; make -C .../csrc)
# against this one, default bench run with the extra legs, in ONE gpurun call: tools/ab_tree.sh <name> [rounds]
cd ${GRAFT_REPO_ROOT:-.}
ROOT=$PWD; name=$1; rounds=${2:-2}
for r in $(seq $rounds); do
  for t in $name HEAD; do
    if [ $t = HEAD ]; then cd $ROOT; else cd $ROOT/tools/ab/$name; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', d['value'], d['ms_per_step'], 'host', d['value_host_entry']['value'], d['value_host_entry_async']['value'], 'fast', d['value_fast_cadence']['value'], 'lc', d['value_close_loops']['value'], 'sharded', d['value_sharded']['value'], d['value_sharded']['ms_per_frame_gpu'], 'call', d['instance']['ms_per_call'])"
  done
done
