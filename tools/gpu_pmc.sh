#!/bin/bash
# PMC passes over the bench (variant $V), one counter group per run (no trace domains mixed in).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${V:-0}
EXTRA=${EXTRA:-}
TAG=${TAG:-v$V}
# the real interpreter binary: nothing that re-execs may sit between rocprofv3's `--` and the program
PYTHON=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $PYTHON $R/bench.py --steps 5 --warmup 40 --variant $V --cpu-epochs 0 --no-live-traffic $EXTRA > $OUT/$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT
run c GRBM_GUI_ACTIVE FETCH_SIZE
run e SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC
run f SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_VALU2
cd $R && python3 tools/pmc_summary.py $OUT
