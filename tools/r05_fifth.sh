#!/bin/bash
# round 5, fifth GPU call: GPU suite, two-process soak on the epoch-tagged exchange, the round's profile
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_fifth; mkdir -p $O
timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_lnx_tagged.jsonl > $O/soak.log 2>&1
grep -c '"equal": false' $O/soak_lnx_tagged.jsonl
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 12 $O/pytest_gpu.log
python bench.py > $O/bench_1gpu.json 2> $O/bench.err
bash tools/profile_round.sh r05 > $O/profile.log 2>&1
tail -n 30 $O/profile.log
