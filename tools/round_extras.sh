#!/bin/bash
# round-end extras on the GPU box: the GPU suite with this round's new defaults switched OFF (the old launches must still agree with the oracle), other map sizes / resolutions
TAG=${1:-r05_x}
cd ${GRAFT_REPO_ROOT:-.}
IFX_OPTS="clean_raster=0,hot_records=0" python -m pytest tests -m gpu -q -k "not config5_8_streams and not config4 and not sweep" 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > gpurun_out/${TAG}_gputests_new_defaults_off.txt
cat gpurun_out/${TAG}_gputests_new_defaults_off.txt
IFX_OPTS="hot_records=0" python -m pytest tests/test_gpu_sweep.py -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 >> gpurun_out/${TAG}_gputests_new_defaults_off.txt
tail -3 gpurun_out/${TAG}_gputests_new_defaults_off.txt
for n in 1000000 50000000; do
  timeout 900 python bench.py --surfels $n --no-cpu-baseline --extras-frames 0 --steps 100 --warmup 30 2>/dev/null | tail -1 | python3 -c "
import sys,json;d=json.loads(sys.stdin.read());print(sys.argv[1],d['value'],d['ms_per_frame_gpu'],d['config']['surfel_slots'])" $n
done | tee gpurun_out/${TAG}_size_sweep.txt
timeout 900 python bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 --res 1280x960 --surfels 20000000 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_1280x960_20M.json
python3 -c "
import json;d=json.loads(open('gpurun_out/${TAG}_bench_1280x960_20M.json').read());print('1280x960 20M', d['value'], d['ms_per_frame_gpu'], d.get('exact_sum_range_exceeded'))" | tee -a gpurun_out/${TAG}_size_sweep.txt
