#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats + separate PMC passes over bench.py.
# Run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01'
# Counters are collected in their own runs with --kernel-trace only.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 34 --warmup 4 --min-seconds 0 --no-cpu-baseline --no-extra"   # 34 x 1024 images = 4 tower passes of 8704
# (1) the opt-in two-lane pipeline (LLA_VIT_STREAMS=2): kernels of the two lanes overlap, so their traced durations
#     include the time they share the chip; kept for the record (kernel_stats_two_streams.csv)
# (retired in round 4: the product library has one tower stream)
# LLA_VIT_STREAMS=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o ${TAG}_two_streams -- $BENCH > $OUT/bench_under_trace_two_streams.json 2> $OUT/trace2.err
# (2) the default command (one in-order tower stream since round 3): a kernel's duration and counters are its own
#     (this is what bench.py's `roofline` object measures with HIP events)
export LLA_VIT_STREAMS=1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.err
pmc() { name=$1; shift
  # (counter passes: only full tower passes of 8704 images, so that "per launch" means the same launch mix as bench.py's
  #  roofline object: no warm-up slice, no 1024-image verification passes)
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BENCH --warmup 0 --no-verify --no-profile > $OUT/$name.json 2> $OUT/$name.err; }
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
pmc pmc_l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
# (3) RN50-CLIP tower and the hipBLASLt ceiling of the four layer GEMMs on THIS box
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rn50 -o ${TAG}_rn50 -- python $REPO/tools/rn50_bench.py 1024 1024 3 > $OUT/rn50_bench.txt 2> $OUT/rn50.err
timeout 300 python $REPO/tools/blas_ceiling.py > $OUT/hipblaslt_ceiling.txt 2>&1
timeout 300 python $REPO/tools/gemm_bench.py 51200 40 > $OUT/gemm_bench.txt 2>&1
cd $REPO
python tools/profile_summary.py $OUT $TAG > $OUT/summary.txt 2>&1
python tools/profile_summary.py $OUT $TAG rn50 >> $OUT/summary.txt 2>&1
python tools/profile_summary.py $OUT $TAG trace2 >> $OUT/summary.txt 2>&1
ls $OUT
