#!/bin/bash
# Partial refresh of a round's profile after a change that leaves the ViT kernels alone: kernel trace of the default
# bench command, the RN50 trace, and the full bench line.  (tools/profile_round.sh is the whole thing incl. PMC passes.)
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace $OUT/rn50
BENCH="python $REPO/bench.py --steps 34 --warmup 4 --min-seconds 0 --no-cpu-baseline --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rn50 -o ${TAG}_rn50 -- python $REPO/tools/rn50_bench.py 1024 1024 3 > $OUT/rn50_bench.txt 2> $OUT/rn50.err
cd $REPO
python - <<PY
import sys
sys.argv = ["profile_summary.py", "$OUT", "$TAG"]
sys.path.insert(0, "tools")
import profile_summary as ps
ps.kernel_stats("$OUT", "trace", "kernel_stats.csv")
ps.kernel_stats("$OUT", "rn50", "kernel_stats_rn50.csv")
PY
timeout 900 python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err
tail -c 600 $OUT/bench_1gpu.err
head -c 300 $OUT/bench_1gpu.json
