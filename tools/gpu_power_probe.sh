#!/bin/bash
# Board power and clocks while a kernel runs back to back (rocm-smi polled from a second process).
probe() {  # label, bench args
  python bench.py --cpu-epochs 0 --no-live-traffic --no-alt --steps 4000 --warmup 50 $2 > /tmp/b.json 2>/dev/null &
  pid=$!
  sleep 4
  for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks --showuse --json 2>/dev/null | python3 -c "
import sys, json
try:
    j = json.load(sys.stdin); c = j.get('card0', {})
    keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'fclk', 'gpu use'))}
    print('$1', keep)
except Exception as e:
    print('$1 smi parse failed', e)"
    sleep 0.5
  done
  wait $pid
  python3 -c "
import json; j = json.load(open('/tmp/b.json')); print('$1 frac', round(j['roofline']['frac'], 4))"
}
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|cap" | head -3
probe headline ""
probe headline_zeros "--zeros"
probe welch "--mode welch"
probe welch_zeros "--mode welch --zeros"
