#!/bin/bash
# round 5, eleventh GPU call: the CU-mask partition loaded before anything starts the runtime -- soak and the sharding tests
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_eleventh; mkdir -p $O
timeout 900 python tools/two_rank_soak.py --runs 20 --out $O/soak_partitioned.jsonl > $O/soak.log 2>&1
grep -c '"equal": false' $O/soak_partitioned.jsonl; tail -n 1 $O/soak_partitioned.jsonl
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_distributed.py -q -m gpu > $O/pytest.log 2>&1
tail -n 8 $O/pytest.log
