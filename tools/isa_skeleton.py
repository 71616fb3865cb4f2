#!/usr/bin/env python3
"""Print the non-VALU skeleton of an AMDGPU .s fragment with VALU counts between entries."""
import sys
v = 0
for line in sys.stdin:
    s = line.strip()
    if not s or s.startswith(';'):
        continue
    op = s.split()[0]
    if op.startswith('v_'):
        v += 1
        continue
    if op.startswith(('ds_', 'buffer_', 'global_', 's_waitcnt', 's_barrier', 's_cbranch', '.LBB', 's_setprio', 'scratch_')):
        if v:
            print('    [%d VALU]' % v)
            v = 0
        print(s[:90])
if v:
    print('    [%d VALU]' % v)
