#!/bin/bash
# round 5, eighth GPU call: HSA_CU_MASK bit -> XCD mapping; the GPU suite and the two-process soak with ranks that share
# the GPU on disjoint XCDs (bench.py's default now); the round's bench line and profile on the final tree
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_eighth; mkdir -p $O
make -C tools/ubench xcc_map > /dev/null 2>&1
( echo "no mask"; tools/ubench/xcc_map; for m in 0-127 128-255 0-31 32-63 224-255; do echo "HSA_CU_MASK=0:$m"; HSA_CU_MASK=0:$m tools/ubench/xcc_map; done ) > $O/cu_mask_xcc_map.txt 2>&1
cat $O/cu_mask_xcc_map.txt
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 8 $O/pytest_gpu.log
timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_disjoint_xcds.jsonl > $O/soak.log 2>&1
grep -c '"equal": false' $O/soak_disjoint_xcds.jsonl; tail -n 1 $O/soak_disjoint_xcds.jsonl
python bench.py > $O/bench_1gpu.json 2> $O/bench.err
bash tools/profile_round.sh r05 > $O/profile.log 2>&1
tail -n 25 $O/profile.log
