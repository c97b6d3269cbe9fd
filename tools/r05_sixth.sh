#!/bin/bash
# round 5, sixth GPU call: what makes two processes on one GPU differ?  The reproducer with back-to-back
# launches; the soak with an agent-scope acquire (and release) of its own in every tower kernel; the soak with the two
# ranks on disjoint CUs / disjoint XCDs (HSA_CU_MASK)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_sixth; mkdir -p $O
V=$PWD/lossyless_amd/variants
B="python bench.py --no-cpu-baseline --no-extra --no-verify"
# A/B on one box, interleaved: LDS-DMA through a buffer descriptor (3 instructions per piece) vs global_load_lds (6-7)
for i in 1 2; do
  $B > $O/bench_bufdma_$i.json 2>> $O/bench.err
  LLA_LIB=$V/liblossyless_amd_nobufdma.so $B > $O/bench_nobufdma_$i.json 2>> $O/bench.err
done
grep -H -o '"value": [0-9.]*\|"gemm_ms_per_step": [0-9.]*' $O/bench_*.json
timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_vit.py -x -q -m gpu > $O/pytest_q4.log 2>&1; tail -n 3 $O/pytest_q4.log
U=tools/ubench/two_proc_stale
( timeout 200 $U 1 2000 256 50; timeout 200 $U 2 2000 256 50; timeout 300 $U 2 1000 2048 50 ) > $O/two_proc_stale_burst.txt 2>&1
cat $O/two_proc_stale_burst.txt
LLA_LIB=$V/liblossyless_amd_acq1.so timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_acq1.jsonl > $O/soak_acq1.log 2>&1
LLA_LIB=$V/liblossyless_amd_acqrel.so timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_acqrel.jsonl > $O/soak_acqrel.log 2>&1
timeout 900 python tools/two_rank_soak.py --runs 16 --cu-split cu --out $O/soak_split_cu.jsonl > $O/soak_split_cu.log 2>&1
timeout 900 python tools/two_rank_soak.py --runs 16 --cu-split xcd --out $O/soak_split_xcd.jsonl > $O/soak_split_xcd.log 2>&1
for f in acq1 acqrel split_cu split_xcd; do echo $f; grep -c '"equal": false' $O/soak_$f.jsonl; grep -o '"img_per_sec": [0-9.]*' $O/soak_$f.jsonl | head -2; done
