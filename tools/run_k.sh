cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "other_resolutions or gn_prologue" 2>&1 | grep -a "passed\|failed\|^FAILED\|^E  " | head
bash tools/prof_run.sh r04_big --res 1280x960 > /dev/null 2>&1; cat gpurun_out/r04_big_seg_call_timeline.txt | head -70
python bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 --res 1280x960 --surfels 20000000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('1280x960/20M auto', d['value'], d['ms_per_frame_gpu'], d['instance']['ms_per_call'], {n:round(v['avg_ms']*1000,1) for n,v in k.items() if n in ('icp_residual','rgb_step_solve','so3_fused')})"
