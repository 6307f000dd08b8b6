#!/bin/bash
# quick SQ counters of the timed region and of the full-work FAST: bash tools/pmc_quick.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T=${1:-q}
O=gpurun_out/pmcq_$T
mkdir -p $O
C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
rocprofv3 --kernel-trace --pmc $C -d $O/timed -o run --output-format csv -- python3 bench.py --only-timed --steps 3 --warmup 1 --no-cpu-baseline > $O/timed.log 2>&1
rocprofv3 --kernel-trace --pmc $C -d $O/full -o run --output-format csv -- python3 bench.py --only-timed --steps 3 --warmup 1 --no-cpu-baseline --full-work > $O/full.log 2>&1
python3 tools/pmc_summary.py "$O/timed/*counter_collection.csv" "$O/full/*counter_collection.csv" > $O/summary.txt
cat $O/summary.txt
