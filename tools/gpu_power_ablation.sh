#!/bin/bash
export CRN_SENSE_AB=1   # measurement variants are compiled into libcrnsense_ab.so only
# Where the 1400 W go: package power and sclk for the ablation variants of the 4096-point kernel
# (11 stream only, 14 butterflies only, 15 butterflies + LDS exchanges without reload, 16 no epoch close, 0 default).
probe() {
  python bench.py --cpu-epochs 0 --no-live-traffic --no-alt --no-check --steps 3000 --warmup 50 $2 > /tmp/b.json 2>/dev/null &
  pid=$!
  sleep 3.5
  for i in 1 2 3; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys, json
c = json.load(sys.stdin).get('card0', {})
print('$1', 'sclk', c.get('sclk clock speed:'), 'power', c.get('Current Socket Graphics Package Power (W)'))"
    sleep 0.4
  done
  wait $pid
  python3 -c "
import json; j = json.load(open('/tmp/b.json')); r = j['roofline']; print('$1 frac %.4f median ms %.4f' % (r['frac'], r['kernel_ms_median']))"
}
probe default ""
probe stream_only "--variant 11"
probe butterflies_only "--variant 14"
probe compute_no_reload "--variant 15"
probe stream_only_zeros "--variant 11 --zeros"
probe butterflies_only_zeros "--variant 14 --zeros"
