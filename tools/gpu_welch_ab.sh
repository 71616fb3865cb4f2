#!/bin/bash
export CRN_SENSE_AB=1   # measurement variants are compiled into libcrnsense_ab.so only
# A/B of the windowed-kernel variants on one box (boxes differ by a few %: only numbers of one call compare).
for rep in 1 2; do
for v in ${VARIANTS:-22 19 20 21 0}; do
  python bench.py --mode welch --cpu-epochs 0 --no-live-traffic --variant $v ${EXTRA:-} 2>/tmp/err_$v | python -c "
import sys, json
t = sys.stdin.read().strip()
if not t:
    print('variant $v FAILED:', open('/tmp/err_$v').read()[-300:])
else:
    j = json.loads(t); r = j['roofline']; print('variant $v', round(r['frac'], 4), 'median ms', round(r['kernel_ms_median'], 4), 'min', round(r['kernel_ms_min'], 4))"
done; done
