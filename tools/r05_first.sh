#!/bin/bash
# round 5, first GPU call: baseline line, library-slice sweep (does a MALL-resident slice pay?), 1-rank vs 2-rank soak
# with the default library and with write-through (`sc1`) epilogue stores
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_first; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extra --no-verify"
$B > $O/bench_default.json 2> $O/bench_default.err
for c in 384 512 768 1280 2176; do $B --chunk $c > $O/bench_chunk$c.json 2>> $O/bench_default.err; done
LLA_LIB=$PWD/lossyless_amd/variants/liblossyless_amd_stsc1.so $B > $O/bench_stsc1.json 2>> $O/bench_default.err
timeout 900 python tools/two_rank_soak.py --runs 8 --out $O/soak_default.jsonl > $O/soak_default.log 2>&1
LLA_LIB=$PWD/lossyless_amd/variants/liblossyless_amd_stsc1.so timeout 900 python tools/two_rank_soak.py --runs 8 --out $O/soak_stsc1.jsonl > $O/soak_stsc1.log 2>&1
grep -h -o '"value": [0-9.]*' $O/bench_*.json
tail -3 $O/soak_default.jsonl $O/soak_stsc1.jsonl
