#!/bin/bash
# Same binary, same launches, random vs all-zero IQ, interleaved: what the clock the chip holds under load is worth.
for rep in 1 2; do
for args in "--mode welch" "--mode welch --zeros" "" "--zeros" "--fft 2048" "--fft 2048 --zeros" "--mode ref" "--mode ref --zeros"; do
  python bench.py --cpu-epochs 0 --no-live-traffic --no-alt $args 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('%-40s frac %.4f  median ms %.4f  min %.4f' % ('$args', r['frac'], r['kernel_ms_median'], r['kernel_ms_min']))"
done; done
