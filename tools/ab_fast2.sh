#!/bin/bash
# same-box A/B of the two FAST kernels of the whole path, production shape (pipelined) and one batch at a time
mkdir -p gpurun_out/r03
for rep in 1 2; do
for impl in 3 4; do
  ORBX_FAST_IMPL=$impl timeout -k 10 200 python bench.py --only-timed --no-cpu-baseline > gpurun_out/r03/ab.json 2> gpurun_out/r03/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/r03/ab.json')); print('impl', $impl, 'one batch at a time', round(d['value']), d['roofline_kernels_ms'])"
  ORBX_FAST_IMPL=$impl timeout -k 10 300 python bench.py --no-cpu-baseline --strong-frames 0 > gpurun_out/r03/ab.json 2> gpurun_out/r03/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/r03/ab.json')); print('impl', $impl, 'pipelined', round(d['value']), round(d['value_full_work']), round(d['value_sustained']), round(d['fps_with_d2h']))"
  ORBX_FAST_IMPL=$impl timeout -k 10 300 python bench.py --no-cpu-baseline --strong-frames 0 --batch 256 > gpurun_out/r03/ab.json 2> gpurun_out/r03/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/r03/ab.json')); print('impl', $impl, 'pipelined 256', round(d['value']), round(d['value_full_work']), round(d['value_sustained']), round(d['fps_with_d2h']))"
done
done
