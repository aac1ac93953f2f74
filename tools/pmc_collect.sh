#!/bin/bash
# HBM traffic per launch of every kernel of the benchmark, on the GPU box:  tools/pmc_collect.sh <tag> [bench.py args]
# Two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: they do not fit the TCC counter slots together; --pmc is never combined with
# --stats / trace domains other than the kernel trace), then tools/pmc_summary.py -> gpurun_out/<tag>_pmc_traffic.json, which
# records the workload it was measured on (bench.py only uses a file that matches its own workload and kernel set).
TAG=${1:-r02_x}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="--steps 40 --warmup 10 --no-cpu-baseline --extras-frames 0 $@"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_$TAG -o f -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmc_bench.json 2> gpurun_out/${TAG}_pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write_$TAG -o w -- python3 bench.py $ARGS > /dev/null 2> gpurun_out/${TAG}_pmc_w.err
F=$(find gpurun_out/pmc_fetch_$TAG -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_write_$TAG -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py pmc "$F" "$W" gpurun_out/${TAG}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on python3 bench.py $ARGS" gpurun_out/${TAG}_pmc_bench.json "$(ls gpurun_out/*_pmc_calibration.json profiles/*_pmc_calibration.json 2>/dev/null | tail -1)"
rm -rf gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG
python3 - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_pmc_traffic.json"))
print(j.get("workload"))
for k,v in list(j["kernels"].items())[:14]: print(k, v)
PY
