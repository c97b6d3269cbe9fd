#!/bin/bash
# round 5, after the triple walk went in: the 1-rank vs 2-rank (two processes on the one GPU, contiguous halves of the CU mask)
# file identity soak again on the final tree -- 20 runs of 10^6 images (profiles/r05_two_rank_soak.txt, last row)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_soak_final; mkdir -p $O
timeout 1100 python tools/two_rank_soak.py --runs 20 --images 1000000 --out $O/soak.jsonl > $O/soak.log 2>&1
echo "exit (= mismatching runs): $?"
tail -n 3 $O/soak.log
