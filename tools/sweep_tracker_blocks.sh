#!/bin/bash
# frame rate / tracker launch times against the block counts of the two tracker launches (run on the GPU box)
for o in "" "--opt icp_blocks=456" "--opt icp_blocks=608" "--opt icp_blocks=1200" "--opt res_blocks=300" "--opt res_blocks=600" "--opt icp_blocks=608 --opt res_blocks=600" "--opt icp_blocks=1200 --opt res_blocks=600" "--opt rgb_blocks=96" "--opt rgb_blocks=300"; do
  python bench.py --gpus 1 --steps 150 --warmup 30 --no-cpu-baseline --extras-frames 0 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$o', d['value'], d['ms_per_frame_gpu']['track'], 'icp', k['icp_residual']['avg_ms'], 'rgb', k['rgb_step_solve']['avg_ms'])"
done
