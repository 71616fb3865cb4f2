#!/bin/bash
export CRN_SENSE_AB=1   # measurement variants are compiled into libcrnsense_ab.so only
# First GPU session: parity tests, smoke, bench per kernel variant.
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
for v in 1 2 3 4 5 6 7 8; do
  timeout 300 python bench.py --steps 10 --warmup 2 --variant $v --cpu-epochs 0 > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err
done
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench_v*.json gpurun_out/bench_default.json
