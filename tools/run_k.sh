cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "persistent or configurations or resolutions" > gpurun_out/r03_y_tests.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r03_y_tests.log | head -5
for r in 1 2; do for a in "gn_persist_blocks=128" "gn_persist_blocks=100000"; do python bench.py --steps 100 --warmup 20 --no-cpu-baseline --extras-frames 0 --res 1280x960 --surfels 20000000 --opt $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1280x960 $a', d['value'], d['ms_per_frame_gpu'])"; done; done
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --extras-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_frame_gpu'], r['kernel'], r['frac'], r['traffic'], [ (m['kernel'], m['traffic_over_algorithmic']) for m in r['map_passes']][:4])"
