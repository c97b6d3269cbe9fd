#!/bin/bash
# round 5, final check on the committed tree: smoke(), the whole GPU suite as the driver runs it, the bench line
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 2 $O/smoke.log
timeout 2700 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -n 5 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench.err ) 2>&1 | tail -3
grep -o '"value": [0-9.]*\|"verified": [a-z]*' $O/bench_driver_style.json | head -3
