#!/bin/bash
# Round 3, third GPU call: GPU suite; Welch with the long-span geometry: bench lines (K = 8, K = 32), HBM traffic passes, finer span sweep.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
CRN_EVIDENCE_DIR=$O timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout 300 python bench.py --mode welch --cpu-epochs 0 > $O/bench_cfg2_welch.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --mode welch --cpu-epochs 0 --frames 32 > $O/bench_cfg2_welch_K32.json 2> $O/bench_cfg2_K32.err
python - <<'PY'
import json
for f in ("bench_cfg2_welch", "bench_cfg2_welch_K32"):
    d=json.load(open(f"gpurun_out/r03c/{f}.json")); r=d["roofline"]
    print(f, round(r["frac"],4), r["kernel_ms_mean"], r["traffic"], r["traffic"]/d["config"]["bytes_per_gpu_per_step"] if r["traffic"] else None)
PY
