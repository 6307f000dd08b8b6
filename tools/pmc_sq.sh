#!/bin/bash
# SQ counters per kernel (two passes); run on the GPU box: bash tools/pmc_sq.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/pmcA -o run --output-format csv -- python3 bench.py --only-timed --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d gpurun_out/pmcB -o run --output-format csv -- python3 bench.py --only-timed --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcB.log 2>&1
python3 tools/pmc_summary.py "gpurun_out/pmcA/*counter_collection.csv" "gpurun_out/pmcB/*counter_collection.csv" > gpurun_out/pmc_sq_now.txt
