#!/bin/bash
# round 5, second GPU call: the reworked bench line, the new / changed GPU tests, and the 1-rank vs 2-rank soak with
# EVERY activation load past the vector L1 (`sc1` on the GEMMs' LDS-DMA operand loads, the residual rows, attention's qkv)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_second; mkdir -p $O
python bench.py --no-cpu-baseline --no-extra > $O/bench_default.json 2> $O/bench_default.err
timeout 1500 python -m pytest tests/test_gpu_pass_size.py tests/test_gpu_compressor.py tests/test_gpu_distributed.py "tests/test_preprocess.py::test_array_backed_datasets_are_read_from_their_array_and_write_the_same_file" -x -q -m gpu > $O/pytest.log 2>&1
V=$PWD/lossyless_amd/variants/liblossyless_amd_ldsc1.so
LLA_LIB=$V python bench.py --no-cpu-baseline --no-extra --no-verify > $O/bench_ldsc1.json 2>> $O/bench_default.err
LLA_LIB=$V timeout 1200 python tools/two_rank_soak.py --runs 24 --out $O/soak_ldsc1.jsonl > $O/soak_ldsc1.log 2>&1
grep -h -o '"value": [0-9.]*' $O/bench_*.json
tail -n 5 $O/pytest.log
tail -n 2 $O/soak_ldsc1.jsonl | cut -c1-400
