#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the map passes (run on the GPU box): tools/micro/pmc_calib under the counters,
# known byte counts against the counter -> gpurun_out/<tag>_pmc_calibration.json (factor per pattern; tools/pmc_summary.py applies them)
TAG=${1:-r03_x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_cal_$TAG -o c -- tools/micro/pmc_calib > gpurun_out/${TAG}_pmc_calib_expected.json 2> gpurun_out/${TAG}_pmc_calib.err
F=$(find gpurun_out/pmc_cal_$TAG -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py calib "$F" gpurun_out/${TAG}_pmc_calib_expected.json gpurun_out/${TAG}_pmc_calibration.json
rm -rf gpurun_out/pmc_cal_$TAG
cat gpurun_out/${TAG}_pmc_calibration.json
