#!/bin/bash
# FETCH_SIZE calibration for this kernel's access width (8 B per lane): known byte count per launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/calib; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/membw -- $R/tools/membw > $O/membw.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/membw/*/*_counter_collection.csv")[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='FETCH_SIZE': acc[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, "FETCH_SIZE mean KB=%.0f -> bytes=%.4g ; known bytes per launch=2348810240 ; ratio known/reported=%.3f"%(sum(v)/len(v), sum(v)/len(v)*1024, 2348810240/(sum(v)/len(v)*1024)))
PY
