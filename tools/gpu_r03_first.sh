#!/bin/bash
# Round 3, first GPU call: the whole GPU suite (new: engine ce_args, 8-rank self-launched bench, per-bin error vs SNR), smoke,
# the driver-shaped bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
CRN_EVIDENCE_DIR=$O timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
tail -15 $O/pytest_gpu.log; tail -3 $O/smoke.log; cut -c1-600 $O/bench_driver_shape.json
