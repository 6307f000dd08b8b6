#!/bin/bash
# HBM traffic counters per kernel, separate passes (run on the GPU box: bash tools/pmc_traffic.sh)
set -e
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python3 bench.py --only-timed --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcF -o run --output-format csv -- $B > gpurun_out/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcW -o run --output-format csv -- $B > gpurun_out/pmcW.log 2>&1
export ORBX_FAST_EARLY=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcF0 -o run --output-format csv -- $B > gpurun_out/pmcF0.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmcW0 -o run --output-format csv -- $B > gpurun_out/pmcW0.log 2>&1
unset ORBX_FAST_EARLY
python3 tools/pmc_summary.py "gpurun_out/pmcF/*counter_collection.csv" "gpurun_out/pmcW/*counter_collection.csv" "gpurun_out/pmcF0/*counter_collection.csv" "gpurun_out/pmcW0/*counter_collection.csv" > gpurun_out/pmc_traffic_now.txt
