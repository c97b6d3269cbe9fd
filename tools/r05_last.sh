#!/bin/bash
# round 5, last GPU call on the committed tree: the GPU suite as the driver runs it, the bench line, the round's profile
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_last; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -n 4 $O/pytest_gpu.log
python bench.py > $O/bench_1gpu.json 2> $O/bench.err
grep -o '"value": [0-9.]*\|"verified": [a-z]*' $O/bench_1gpu.json | head -2
bash tools/profile_round.sh r05 > $O/profile.log 2>&1
tail -n 6 $O/profile.log
