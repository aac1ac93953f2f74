#!/bin/bash
# kernel trace of the benchmark on the GPU box: gpurun_out/<tag>_kernel_stats.csv + the bench line printed under the profiler + the timeline of one segmentation call
TAG=${1:-r03_x}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --steps 90 --warmup 20 --no-cpu-baseline --extras-frames 0 "$@" > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.err
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 tools/pmc_summary.py stats "$f" gpurun_out/${TAG}_kernel_stats.csv
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 tools/seg_timeline.py "$t" > gpurun_out/${TAG}_seg_call_timeline.txt 2>&1
python3 tools/frame_timeline.py "$t" > gpurun_out/${TAG}_frame_timeline.txt 2>&1
gzip -c "$t" > gpurun_out/${TAG}_kernel_trace.csv.gz
rm -rf gpurun_out/prof_$TAG
head -30 gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_seg_call_timeline.txt
