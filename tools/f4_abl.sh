mkdir -p gpurun_out/r03
for a in 0 8 1 2; do
  ORBX_F4_ABL=$a timeout -k 10 200 python bench.py --only-timed --full-work --no-cpu-baseline > gpurun_out/r03/abl_$a.json 2> gpurun_out/r03/abl_$a.err
  python -c "
import json; d=json.load(open('gpurun_out/r03/abl_$a.json')); print('abl', $a, d['value'], d['roofline_kernels_ms'])"
done
