#!/bin/bash
# Round-4 GPU call B: the whole GPU suite (no -x), smoke, the decision band with the Welch plan.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
CRN_EVIDENCE_DIR=$O timeout 3000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
tail -40 $O/pytest_gpu.log; tail -8 $O/smoke.log; cat $O/decision_band.txt
