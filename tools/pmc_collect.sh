#!/bin/bash
# rocprofv3 evidence for profiles/r03 (run on the GPU box: bash tools/pmc_collect.sh).
#   kernel-trace --stats and the PMC passes are SEPARATE runs (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2:
#   /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"); the program after `--` is python3 itself.
# Configurations of bench.py (--only-timed: nothing but the timed region runs, so the per-kernel averages
# describe exactly that configuration):
#   timed      production: pyramid+blur fused, FAST (k_fast4) early exit on
#   fullwork   FAST with every tile working (--full-work)
#   unfused    separate pyramid and blur kernels (--unfused)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03c
mkdir -p $O
B="bench.py --only-timed --no-cpu-baseline --steps 6 --warmup 4"  # (4 warm-up steps: the adaptive first pass re-partitions the tile rows after the second batch; pmc_to_json.py leaves them out)   # bench.py's default workload: 1024 frames per launch
B64="bench.py --only-timed --no-cpu-baseline --batch 64 --rotate 4 --steps 10 --warmup 4"  # BASELINE.json configs[2]
run() {  # name, rocprof args..., -- bench args
  local name=$1; shift
  echo "== $name: $*" >> $O/log.txt
  rocprofv3 "$@" >> $O/log.txt 2>&1 || echo "FAILED $name" >> $O/log.txt
  echo "$name done"
}
SQA="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQB="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"
for cfg in timed fullwork unfused; do
  case $cfg in
    timed) X="";;
    fullwork) X="--full-work";;
    unfused) X="--unfused";;
  esac
  run ${cfg}_stats --kernel-trace --stats -d $O/${cfg}_stats -o run --output-format csv -- python3 $B $X
  run ${cfg}_sqa --kernel-trace --pmc $SQA -d $O/${cfg}_sqa -o run --output-format csv -- python3 $B $X
  run ${cfg}_sqb --kernel-trace --pmc $SQB -d $O/${cfg}_sqb -o run --output-format csv -- python3 $B $X
  run ${cfg}_fetch --kernel-trace --pmc FETCH_SIZE -d $O/${cfg}_fetch -o run --output-format csv -- python3 $B $X
  run ${cfg}_write --kernel-trace --pmc WRITE_SIZE -d $O/${cfg}_write -o run --output-format csv -- python3 $B $X
done
# the LDS tile FAST kernel (ORBX_FAST_IMPL=3; the default whole path runs the register-streaming kernel), every tile
# working, and in production
export ORBX_FAST_IMPL=3
run fast3_stats --kernel-trace --stats -d $O/fast3_stats -o run --output-format csv -- python3 $B --full-work
run fast3_sqa --kernel-trace --pmc $SQA -d $O/fast3_sqa -o run --output-format csv -- python3 $B --full-work
run fast3_sqb --kernel-trace --pmc $SQB -d $O/fast3_sqb -o run --output-format csv -- python3 $B --full-work
run fast3_fetch --kernel-trace --pmc FETCH_SIZE -d $O/fast3_fetch -o run --output-format csv -- python3 $B --full-work
run fast3_write --kernel-trace --pmc WRITE_SIZE -d $O/fast3_write -o run --output-format csv -- python3 $B --full-work
run fast3timed_stats --kernel-trace --stats -d $O/fast3timed_stats -o run --output-format csv -- python3 $B
unset ORBX_FAST_IMPL
# kernel durations at 64 frames per launch (BASELINE.json configs[2]; pools inside the Infinity Cache) and at 256
B256="bench.py --only-timed --no-cpu-baseline --batch 256 --rotate 2 --steps 10 --warmup 4"
for cfg in timed fullwork unfused; do
  case $cfg in
    timed) X="";;
    fullwork) X="--full-work";;
    unfused) X="--unfused";;
  esac
  run b64_${cfg}_stats --kernel-trace --stats -d $O/b64_${cfg}_stats -o run --output-format csv -- python3 $B64 $X
  run b256_${cfg}_stats --kernel-trace --stats -d $O/b256_${cfg}_stats -o run --output-format csv -- python3 $B256 $X
done
# BASELINE.json configs[4]: 1920x1080, 12 levels, 4000 features
run hd_timed_stats --kernel-trace --stats -d $O/hd_timed_stats -o run --output-format csv -- python3 bench.py --workload 1080p --batch 32 --rotate 2 --steps 6 --warmup 4 --only-timed --no-cpu-baseline
run hd_fullwork_stats --kernel-trace --stats -d $O/hd_fullwork_stats -o run --output-format csv -- python3 bench.py --workload 1080p --batch 32 --rotate 2 --steps 6 --warmup 4 --only-timed --no-cpu-baseline --full-work
# plain bench lines (no profiler): the default line, configs[2] (64 frames per step), configs[3] at N = 1
# (8 x 1000-frame stream walked once), configs[4]
python3 bench.py > $O/bench_default.json 2>> $O/log.txt || echo "FAILED bench_default" >> $O/log.txt
python3 bench.py --batch 64 --rotate 4 --no-cpu-baseline > $O/bench_batch64.json 2>> $O/log.txt || echo "FAILED bench_batch64" >> $O/log.txt
ORBX_FAST_IMPL=3 python3 bench.py --no-cpu-baseline --strong-frames 0 > $O/bench_fast3.json 2>> $O/log.txt || echo "FAILED bench_fast3" >> $O/log.txt
python3 bench.py --workload 1080p --batch 32 --rotate 2 --steps 10 --no-cpu-baseline > $O/bench_1080p.json 2>> $O/log.txt || echo "FAILED bench_1080p" >> $O/log.txt
echo "bench lines done"
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts, per access width (tools/bw_probe.hip)
[ -x tools/bw_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/bw_probe.bin tools/bw_probe.hip >> $O/log.txt 2>&1 || true
if [ -x tools/bw_probe.bin ] && [ -z "$SKIP_CALIB" ]; then  # (SKIP_CALIB=1: the probes do not depend on the library; keep the last calibration)
  for mb in 97 1600; do
    ./tools/bw_probe.bin $mb > $O/bw_probe_$mb.txt 2>&1
    run calib_fetch_$mb --kernel-trace --pmc FETCH_SIZE -d $O/calib_fetch_$mb -o run --output-format csv -- ./tools/bw_probe.bin $mb
    run calib_write_$mb --kernel-trace --pmc WRITE_SIZE -d $O/calib_write_$mb -o run --output-format csv -- ./tools/bw_probe.bin $mb
  done
fi
python3 tools/pmc_to_json.py $O profiles/r03 1024 4 > $O/summary.txt 2>&1 || true
mkdir -p $O/profiles_r03 && cp profiles/r03/*.csv profiles/r03/*.json profiles/r03/*.txt $O/profiles_r03/ 2>/dev/null || true
tail -5 $O/summary.txt
