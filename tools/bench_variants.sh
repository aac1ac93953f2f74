for a in "x" "--no-prefetch" "--opt two_streams=0"; do
  [ "$a" = "x" ] && a=""
  timeout 300 python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_frame_gpu'], d['ate_rms_m'])" $a
done
