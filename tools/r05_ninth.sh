#!/bin/bash
# round 5, ninth GPU call: LayerNorm / clean-up loads plain again -- two-process soaks without a CU mask, with the
# range mask bench.py now sets, with the comma-list mask of the seventh call; what a mask does to the CU count
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_ninth; mkdir -p $O
P='import torch; p=torch.cuda.get_device_properties(0); print(p.multi_processor_count)'
L=$(python -c 'print(",".join(str(i) for i in range(128)))')
( echo -n "no mask: "; python -c "$P"; echo -n "0:0-127: "; HSA_CU_MASK=0:0-127 python -c "$P"; echo -n "0:<list 0..127>: "; HSA_CU_MASK=0:$L python -c "$P"; echo -n "0:0-31: "; HSA_CU_MASK=0:0-31 python -c "$P" ) > $O/cu_count.txt 2>&1
cat $O/cu_count.txt
timeout 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_bench_config.py -x -q -m gpu > $O/pytest.log 2>&1; tail -n 2 $O/pytest.log
timeout 600 python tools/two_rank_soak.py --runs 12 --cu-split none --out $O/soak_plain_ln_shared.jsonl > $O/s1.log 2>&1
timeout 600 python tools/two_rank_soak.py --runs 8 --out $O/soak_plain_ln_range_mask.jsonl > $O/s2.log 2>&1
timeout 600 python tools/two_rank_soak.py --runs 8 --cu-split cu --out $O/soak_plain_ln_list_mask.jsonl > $O/s3.log 2>&1
for f in shared range_mask list_mask; do echo $f; grep -c '"equal": false' $O/soak_plain_ln_$f.jsonl; grep -o '"differing_records": [0-9]*' $O/soak_plain_ln_$f.jsonl | sort | uniq -c | head -6; done
