for n in 1000000 2000000 5000000 10000000 20000000 50000000; do
  timeout 900 python bench.py --surfels $n --no-cpu-baseline --extras-frames 0 --steps 150 --warmup 30 2>/dev/null | tail -1 | python3 -c "
import sys,json;d=json.loads(sys.stdin.read());print(sys.argv[1],d['value'],d['ms_per_frame_gpu'],d['config']['surfel_slots'])" $n
done
