#!/bin/bash
# round 5, fourth GPU call: the whole GPU suite on the cleaned library, then the 1-rank vs 2-rank soak with LayerNorm in the GEMM epilogues
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_fourth; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 15 $O/pytest_gpu.log
timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_lnx.jsonl > $O/soak_lnx.log 2>&1
tail -n 1 $O/soak_lnx.jsonl
grep -c '"equal": false' $O/soak_lnx.jsonl
