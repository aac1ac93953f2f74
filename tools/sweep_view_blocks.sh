#!/bin/bash
# map-stage launch times against the grids of the view-list kernels (run on the GPU box)
for o in "--opt view_blocks=4096" "--opt view_blocks=8192" "--opt view_blocks=16384" "--opt view_blocks=4096 --opt clean_blocks=2048" "--opt view_blocks=4096 --opt clean_blocks=4096" "--opt view_blocks=8192 --opt clean_blocks=8192"; do
  python bench.py --gpus 1 --steps 150 --warmup 30 --no-cpu-baseline --extras-frames 0 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$o', d['value'], d['ms_per_frame_gpu']['fuse'], 'index', k['index_list']['avg_ms'], 'clean', k['clean_view']['avg_ms'], 'raster', k['raster_view']['avg_ms'])"
done
