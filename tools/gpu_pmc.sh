#!/bin/bash
# PMC passes over the bench (variant $V), one counter group per run (no trace domains mixed in).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${V:-7}
OUT=$R/gpurun_out/pmc_v$V
mkdir -p $OUT
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 3 --warmup 1 --variant $V --cpu-epochs 0 > $OUT/$name.log 2>&1
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run c GRBM_GUI_ACTIVE FETCH_SIZE
run d GRBM_GUI_ACTIVE WRITE_SIZE TCC_HIT_sum
run e SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC
find $OUT -name "*.csv" | head -30
