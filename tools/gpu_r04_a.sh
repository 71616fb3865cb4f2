#!/bin/bash
# Round-4 GPU call A: GPU tests (new ABI, calibration on the launcher thread, decision band), the Welch LDS-traffic A/B with counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
CRN_SENSE_AB=1 timeout 600 python tests/ab_variants_check.py > $O/ab_variants_check.log 2>&1; echo "exit $?" >> $O/ab_variants_check.log
CRN_EVIDENCE_DIR=$O timeout 1200 python -m pytest tests/test_decision_band.py -m gpu -q -x -s > $O/decision_band.log 2>&1; echo "exit $?" >> $O/decision_band.log
VARIANTS="0 25 26 27" bash tools/gpu_welch_ab.sh > $O/welch_ab.txt 2>&1
VARIANTS="0 25 26 27" EXTRA="--frames 32" bash tools/gpu_welch_ab.sh > $O/welch_ab_K32.txt 2>&1
for v in 0 25 26; do V=$v EXTRA="--mode welch --no-alt" TAG=r04_welch_v$v bash tools/gpu_pmc.sh > $O/pmc_welch_v$v.txt 2>&1; done
CRN_EVIDENCE_DIR=$O timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/ab_variants_check.log; tail -30 $O/decision_band.log; cat $O/welch_ab.txt $O/welch_ab_K32.txt; tail -15 $O/pytest_gpu.log
