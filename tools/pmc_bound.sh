#!/bin/bash
# What bounds the list-driven map passes?  tools/pmc_bound.sh <tag> [bench.py args]   (on the GPU box)
# Separate rocprofv3 --pmc passes (never combined with --stats or a trace domain other than the kernel trace; the program itself after `--`):
# SQ occupancy / wait / issue, SQ instruction mix, TCP (vector L1) requests and stalls, TCC (L2) hits / misses / atomics, memory-side atomics.
# A pass whose counters do not fit the block's slots fails on its own and is skipped.  -> gpurun_out/<tag>_pmc_bound.json (tools/pmc_summary.py counters)
TAG=${1:-r04_a}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="--steps 30 --warmup 10 --no-cpu-baseline --extras-frames 0 $@"
i=0
for CTRS in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
  "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum" \
  "TCC_EA0_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/pmcb_${TAG}_$i -o p -- python3 bench.py $ARGS > gpurun_out/${TAG}_pmcb_bench_$i.json 2> gpurun_out/${TAG}_pmcb_$i.err || echo "pass $i failed: $CTRS"
done
python3 tools/pmc_summary.py counters gpurun_out/${TAG}_pmc_bound.json $(find gpurun_out/pmcb_${TAG}_* -name "*counter_collection.csv") 
rm -rf gpurun_out/pmcb_${TAG}_*
