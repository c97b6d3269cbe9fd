#!/bin/bash
# MFMA-busy fraction and shader clock: hipBLASLt vs the ping-pong kernel vs the four-wave kernel (and its timing
# ablations), the four layer shapes at M = 217 600, one process each.  usage: bash tools/clock_probe_q4.sh [tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$REPO/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
run() { name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/gemm_bench.py 217600 12 > $OUT/$name.txt 2>&1; }
if [ -z "$SKIP_BASE" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/blas -o blas -- python $REPO/tools/blas_ceiling.py 217600 > $OUT/blas.txt 2>&1
run pp LLA_GEMM_Q4=0
run q4 LLA_GEMM_Q4=1 LLA_Q4_SCHED=${LLA_Q4_SCHED:-1}
run q4_serial_epilogue LLA_GEMM_Q4=1 LLA_Q4_PIPE=0   # (QKV / FC1: the fp16 epilogue between the tiles instead of in the K loop's shadows)
fi
for d in ${Q4_DBGS:-1 3 13}; do run q4_dbg$d LLA_GEMM_Q4=1 LLA_Q4_DBG=$d; done
cd $REPO
python tools/clock_summary.py "$OUT"
