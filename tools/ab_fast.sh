#!/bin/bash
# same-box A/B of the two FAST kernels of the whole path (development aid)
mkdir -p gpurun_out/r03
for rep in 1 2; do
for impl in 3 4; do
  for mode in "" "--full-work"; do
    ORBX_FAST_IMPL=$impl timeout -k 10 200 python bench.py --only-timed --no-cpu-baseline $mode > gpurun_out/r03/ab.json 2> gpurun_out/r03/ab.err
    python -c "
import json; d=json.load(open('gpurun_out/r03/ab.json')); print('impl', $impl, '$mode', round(d['value']), d['roofline_kernels_ms'])"
  done
done
done
