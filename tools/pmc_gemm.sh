#!/bin/bash
# PMC passes over the GEMM micro-benchmark (run on the GPU box via gpurun).
# Counters go in their own runs with --kernel-trace only (gpurun refuses other combos).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
ARGS="${1:-51200} 3"
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py $ARGS > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
find $OUT -name "*.csv" | head -20
