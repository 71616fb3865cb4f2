#!/bin/bash
export CRN_SENSE_AB=1   # measurement variants are compiled into libcrnsense_ab.so only
# bench every kernel variant (short) + streaming-read calibration + gpu tests
mkdir -p gpurun_out
./tools/membw > gpurun_out/membw.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
for v in ${VARIANTS:-1 2 3 4 5 6 7 8 9 10 11 12 13}; do
  timeout 300 python bench.py --steps 10 --warmup 2 --variant $v --cpu-epochs 0 > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err
done
cat gpurun_out/membw.txt; tail -5 gpurun_out/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_v*.json'), key=lambda s:int(s.split('_v')[1].split('.')[0])):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], d['config']['kernel'], "ach=%.0f GB/s frac=%.3f kern_ms=%.3f min=%.3f"%(r['achieved'],r['frac'],r['kernel_ms_mean'],r['kernel_ms_min']))
    except Exception as e:
        print(f,"ERR",open(f.replace('.json','.err')).read()[-300:])
PY
