#!/bin/bash
# Round 3, second GPU call: full GPU suite on the new build; same-box A/B of the library before / after the sqrt(1/2) twiddle change;
# Welch span sweep; default bench line (with the two-stream 2 GiB leg).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
CRN_EVIDENCE_DIR=$O timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
timeout 1500 bash tools/gpu_ab.sh > $O/ab.txt 2>&1; cat $O/ab.txt
true
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03d/bench_driver_shape.json"))
print(d["value"], d["roofline"]["frac"]); print({k:(round(v["frac"],4)) for k,v in d["config"]["alt"].items()})
PY
