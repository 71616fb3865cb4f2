#!/bin/bash
# Watts per streamed TB/s by load form (tools/membw_power.hip), random data unless noted.
for cfg in "2 2" "2 4" "0 2" "0 4" "16 2" "18 2" "2 2 zeros" "2 4 zeros"; do
  ./tools/membw_power $cfg > /tmp/m.txt &
  pid=$!
  sleep 2.2
  rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys, json
c = json.load(sys.stdin).get('card0', {})
print('[$cfg]', 'sclk', c.get('sclk clock speed:'), 'power', c.get('Current Socket Graphics Package Power (W)'), end='  ')"
  wait $pid
  cat /tmp/m.txt
done
