#!/bin/bash
# A/B on one box, interleaved: row tiles walked in triples that never straddle two rounds vs the plain walk
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_triples3; mkdir -p $O
V=$PWD/lossyless_amd/variants
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_bench_config.py tests/test_gpu_pass_size.py tests/test_gpu_compressor.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extra"
for i in 1 2 3; do
  $B > $O/bench_triples_$i.json 2>> $O/bench.err
  LLA_LIB=$V/liblossyless_amd_notriples.so $B > $O/bench_plainwalk_$i.json 2>> $O/bench.err
done
grep -H -o '"value": [0-9.]*\|"gemm_ms_per_step": [0-9.]*\|"verified": [a-z]*' $O/bench_*.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 34 --warmup 4 --min-seconds 0 --no-cpu-baseline --no-extra > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err
cd $GRAFT_REPO_ROOT; python - <<PY
import sys; sys.path.insert(0,"tools")
import profile_summary as ps
ps.kernel_stats("$O","trace","kernel_stats.csv"); ps.roofline_by_kernel("$O","trace","roofline_by_kernel.csv")
print(open("$O/roofline_by_kernel.csv").read())
PY
