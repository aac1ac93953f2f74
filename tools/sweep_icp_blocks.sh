for b in 128 192 256 304 384 512 768; do
  timeout 300 python bench.py --no-cpu-baseline --steps 150 --opt icp_blocks=$b 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print(sys.argv[1], d['value'], d['ms_per_frame_gpu']['track'], 'icp_residual %.1f rgb_step_solve %.1f' % (1000*k['icp_residual']['avg_ms'], 1000*k['rgb_step_solve']['avg_ms']))" $b
done
