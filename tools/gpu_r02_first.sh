#!/bin/bash
# Round 2, first call: baselines at the round's starting build + SQ/LDS/VMEM counter groups for the
# Welch (cfg2), 2048-point and reference-mode kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
for cfgname in "headline:" "welch:--mode welch" "e2048:--fft 2048" "ref:--mode ref" "e1024:--fft 1024"; do
  tag=${cfgname%%:*}; args=${cfgname#*:}
  timeout 300 python bench.py --cpu-epochs 0 --per-launch-events $args > $O/bench_$tag.json 2> $O/bench_$tag.err
done
TAG=welch EXTRA="--mode welch" bash tools/gpu_pmc.sh > $O/pmc_welch.txt 2>&1
TAG=e2048 EXTRA="--fft 2048" bash tools/gpu_pmc.sh > $O/pmc_e2048.txt 2>&1
TAG=ref EXTRA="--mode ref" bash tools/gpu_pmc.sh > $O/pmc_ref.txt 2>&1
TAG=e1024 EXTRA="--fft 1024" bash tools/gpu_pmc.sh > $O/pmc_e1024.txt 2>&1
cat $O/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['config']['workload'][:50], round(j['value']), j['roofline']['frac'], j['roofline']['kernel_ms_min'], j['roofline']['kernel_ms_median'])
"
cat $O/pmc_welch.txt
