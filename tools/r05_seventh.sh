#!/bin/bash
# round 5, seventh GPU call (after the bit-cast fix in the tagged exchange): LayerNorm-epilogue tests, the LDS-DMA A/B, the
# GPU suite, then the two-process soaks: product build, own acquire / acquire + release in every tower kernel, disjoint CUs / XCDs
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_seventh; mkdir -p $O
V=$PWD/lossyless_amd/variants
T="tests/test_gpu_vit.py::test_layernorm_in_the_residual_gemm_epilogues_gives_the_bits_of_the_layernorm_kernel tests/test_gpu_vit.py::test_large_ragged_batches_are_cut_for_the_four_wave_kernel_and_walked_in_both_directions"
timeout 300 python -m pytest $T -q -m gpu > $O/lnx_product.log 2>&1; tail -n 3 $O/lnx_product.log
B="python bench.py --no-cpu-baseline --no-extra"
for i in 1 2; do
  $B > $O/bench_bufdma_$i.json 2>> $O/bench.err
  LLA_LIB=$V/liblossyless_amd_nobufdma.so $B > $O/bench_nobufdma_$i.json 2>> $O/bench.err
done
grep -H -o '"value": [0-9.]*\|"gemm_ms_per_step": [0-9.]*\|"verified": [a-z]*' $O/bench_*.json
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 12 $O/pytest_gpu.log
timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_product.jsonl > $O/soak_product.log 2>&1
LLA_LIB=$V/liblossyless_amd_acq1.so timeout 900 python tools/two_rank_soak.py --runs 24 --out $O/soak_acq1.jsonl > $O/soak_acq1.log 2>&1
LLA_LIB=$V/liblossyless_amd_acqrel.so timeout 900 python tools/two_rank_soak.py --runs 16 --out $O/soak_acqrel.jsonl > $O/soak_acqrel.log 2>&1
timeout 900 python tools/two_rank_soak.py --runs 16 --cu-split cu --out $O/soak_split_cu.jsonl > $O/soak_split_cu.log 2>&1
timeout 900 python tools/two_rank_soak.py --runs 16 --cu-split xcd --out $O/soak_split_xcd.jsonl > $O/soak_split_xcd.log 2>&1
for f in product acq1 acqrel split_cu split_xcd; do echo $f; grep -c '"equal": false' $O/soak_$f.jsonl; grep -o '"differing_records": [0-9]*' $O/soak_$f.jsonl | sort | uniq -c | head -5; grep -o '"img_per_sec": [0-9.]*' $O/soak_$f.jsonl | head -2; done
