#!/bin/bash
# MFMA-busy fraction and shader clock of the eight-wave GEMM (gemm_w8.hip) and its timing ablations against the four-wave
# kernel, QKV and c_fc shapes at M = 217 600, one process each (clocks run lower under the profiler: compare arms of this
# script with each other only).  usage: bash tools/clock_probe_w8.sh [tag] [arms: q4 w8 w8:p0 w8:3 w8:13 w8@variant ...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; shift
ARMS=${@:-q4 w8 w8:p0 w8:3 w8:13}
OUT=$REPO/gpurun_out/clock_w8_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
for arm in $ARMS; do
  name=${arm//[:@]/_}
  head=${arm%%:*}; opt=""; [ "$arm" != "$head" ] && opt=${arm#*:}
  lib=$REPO/lossyless_amd/liblossyless_amd_ablation.so
  case $head in *@*) lib=$REPO/lossyless_amd/variants/liblossyless_amd_${head#*@}.so;; esac
  w8=1; [ "$head" = q4 ] && w8=0
  extra=""; [ "$opt" = p0 ] && extra="LLA_W8_PIPE=0"; [ -n "$opt" ] && [ "$opt" != p0 ] && extra="LLA_W8_DBG=$opt"
  env LLA_LIB=$lib LLA_GEMM_W8=$w8 $extra timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/w8_probe.py child 217600 12 > $OUT/$name.txt 2>&1
done
cd $REPO
python tools/clock_summary.py "$OUT"
