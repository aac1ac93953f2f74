python -m pytest tests -m gpu -x -q -k "lds_staged" 2>&1 | grep -E "passed|failed|assert" | tail -3
for o in "" "--opt icp_lds=1"; do
  python bench.py --gpus 1 --steps 150 --warmup 30 --no-cpu-baseline --extras-frames 0 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$o', d['value'], d['ms_per_frame_gpu']['track'], 'icp_residual (all levels, HIP events)', k['icp_residual']['avg_ms'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in "" "--opt icp_lds=1"; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lds -o l -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --extras-frames 0 $o > /dev/null 2>&1
f=$(find gpurun_out/prof_lds -name "*kernel_stats.csv" | head -1); echo "rocprof $o"; grep "k_icp_residual" $f | cut -c1-160; rm -rf gpurun_out/prof_lds
done
