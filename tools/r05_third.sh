#!/bin/bash
# round 5, third GPU call: the two-process stale-read reproducer; LayerNorm in the residual GEMM epilogues (tests, bench)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_third; mkdir -p $O
U=tools/ubench/two_proc_stale
( timeout 300 $U 1 300; timeout 300 $U 2 300; timeout 300 $U 1 300 256; timeout 300 $U 2 300 256 ) > $O/two_proc_stale.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_vit.py -x -q -m gpu > $O/pytest_vit.log 2>&1
python bench.py --no-cpu-baseline --no-extra > $O/bench_default.json 2> $O/bench.err
timeout 1500 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_pass_size.py tests/test_gpu_compressor.py -x -q -m gpu > $O/pytest_more.log 2>&1
cat $O/two_proc_stale.txt
tail -n 6 $O/pytest_vit.log; tail -n 4 $O/pytest_more.log
grep -h -o '"value": [0-9.]*\|"gemm_ms_per_step": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_default.json
