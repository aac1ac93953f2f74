#!/bin/bash
# A/B runs of experimental builds of libifx.so (build/variants/libifx_<name>.so, selected through IFX_LIB) on the BASELINE workload:
# frames/s, the stage split and the per-kernel averages of the kernels named in $KERNELS.
#   tools/bench_variants.sh base r8 r32      ("base" = the in-tree library)
KERNELS=${KERNELS:-"cull_raster cull_clean index_project raster_list clean_list icp_residual rgb_step_solve"}
for v in "$@"; do
  if [ "$v" = "base" ]; then unset IFX_LIB; else export IFX_LIB=$PWD/build/variants/libifx_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-150} ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print('%-8s %7.1f fps  %s  ' % (sys.argv[1], d['value'], d['ms_per_frame_gpu']) + '  '.join('%s %.1f' % (n, 1000 * k[n]['avg_ms']) for n in sys.argv[2:] if n in k))" $v $KERNELS
done
