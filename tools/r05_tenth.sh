#!/bin/bash
# round 5, tenth GPU call: which XCDs a comma-list CU mask leaves; two-process soak with bench.py's own partition; the GPU
# suite; the round's bench line and profile on the final tree
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_tenth; mkdir -p $O
make -C tools/ubench xcc_map > /dev/null 2>&1
lst() { python -c "print(','.join(str(i) for i in range($1,$2)))"; }
( echo "no mask"; tools/ubench/xcc_map | head -1
  for r in "0 128" "128 256" "0 32" "32 64" "224 256"; do set -- $r; echo "HSA_CU_MASK=0:<every CU $1..$(($2-1))>"; HSA_CU_MASK=0:$(lst $1 $2) tools/ubench/xcc_map | head -1; done
  echo "HSA_CU_MASK=0:0-127 (range syntax)"; HSA_CU_MASK=0:0-127 tools/ubench/xcc_map | head -1 ) > $O/cu_mask_xcc_map.txt 2>&1
cat $O/cu_mask_xcc_map.txt
timeout 900 python tools/two_rank_soak.py --runs 20 --out $O/soak_partitioned.jsonl > $O/soak.log 2>&1
grep -c '"equal": false' $O/soak_partitioned.jsonl; tail -n 1 $O/soak_partitioned.jsonl
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 8 $O/pytest_gpu.log
python bench.py > $O/bench_1gpu.json 2> $O/bench.err
bash tools/profile_round.sh r05 > $O/profile.log 2>&1
tail -n 12 $O/profile.log
