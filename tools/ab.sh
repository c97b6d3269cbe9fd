#!/bin/bash
# One parametrised A/B driver for the GPU box (replaces the one-off tools/r05_*.sh job scripts of round 5).
# Library builds are named as `make variant NAME=x` names them (lossyless_amd/variants/liblossyless_amd_x.so);
# `product` is the default library; `x@VAR=val,VAR2=val2` adds environment variables to that arm only.
#
#   tools/ab.sh soak  OUT [arm ...]   1-rank file vs RUNS (4) two-rank files of IMAGES (1000000) images per arm
#                                     (tools/two_rank_soak.py; CU_SPLIT=none|cu|xcd, default none = ranks share every XCD)
#   tools/ab.sh bench OUT [arm ...]   ROUNDS (3) interleaved bench.py lines per arm (--no-cpu-baseline --no-extra)
#   tools/ab.sh tower OUT [arm ...]   ROUNDS (3) interleaved tower-only passes per arm (tools/lnx_wait_sweep.py)
#   tools/ab.sh suite OUT             the GPU suite + one bench line + the round's profile (tools/profile_round.sh $ROUND)
#
# Everything lands under gpurun_out/OUT; a summary is printed last (gpurun returns the tail of stdout).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mode=$1; out=$2; shift 2
O=gpurun_out/$out; mkdir -p "$O"
V=$PWD/lossyless_amd/variants
arms=("$@"); [ ${#arms[@]} -eq 0 ] && arms=(product)

run_arm() {   # run_arm ARM cmd...: the arm's library / environment, then the command
  local arm=$1; shift
  local lib=${arm%%@*} extra=""
  [ "$arm" != "$lib" ] && extra=${arm#*@}
  (
    if [ "$lib" != product ]; then export LLA_LIB=$V/liblossyless_amd_$lib.so; [ -f "$LLA_LIB" ] || { echo "missing $LLA_LIB"; exit 9; }; fi
    IFS=, read -ra kv <<< "$extra"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
    "$@"
  )
}

case $mode in
  soak)
    for arm in "${arms[@]}"; do
      tag=${arm//[@=,\/]/_}
      run_arm "$arm" timeout ${SOAK_TIMEOUT:-600} python tools/two_rank_soak.py --runs ${RUNS:-4} --images ${IMAGES:-1000000} \
        --cu-split ${CU_SPLIT:-none} --out $O/soak_$tag.jsonl > $O/soak_$tag.log 2>&1
      echo "$arm: exit (= mismatching runs) $?"
    done
    python - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/soak_*.jsonl")):
    recs = [json.loads(l) for l in open(f)]
    runs = [r for r in recs if isinstance(r.get("run"), int)]
    bad = [r for r in runs if not r["equal"]]
    one = [r for r in recs if r.get("run") == "one_rank"]
    print(os.path.basename(f), "runs", len(runs), "mismatching", len(bad), "records differing",
          [r.get("diff", {}).get("n_differing_records", r.get("n_differing")) for r in bad][:8],
          "img/s 1-rank", [round(r["img_per_sec"]) for r in one])
PY
    ;;
  bench)
    B="python bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS:-}"
    for i in $(seq 1 ${ROUNDS:-3}); do
      for arm in "${arms[@]}"; do
        tag=${arm//[@=,\/]/_}
        run_arm "$arm" $B > $O/bench_${tag}_$i.json 2>> $O/bench.err
      done
    done
    grep -H -o '"value": [0-9.]*\|"gemm_ms_per_step": [0-9.]*\|"verified": [a-z]*' $O/bench_*.json
    ;;
  tower)
    for i in $(seq 1 ${ROUNDS:-3}); do
      for arm in "${arms[@]}"; do
        echo "== round $i $arm" >> $O/sweep.txt
        run_arm "$arm" timeout 120 python tools/lnx_wait_sweep.py --waits ${WAITS:-24000} --rounds 2 --passes 8 >> $O/sweep.txt 2>> $O/err.txt
      done
    done
    grep -v "^library" $O/sweep.txt
    ;;
  suite)
    timeout ${SUITE_TIMEOUT:-1500} python -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1
    tail -n 25 $O/pytest_gpu.log
    python bench.py > $O/bench_1gpu.json 2> $O/bench.err; cat $O/bench_1gpu.json
    [ -n "$ROUND" ] && { bash tools/profile_round.sh $ROUND > $O/profile.log 2>&1; tail -n 30 $O/profile.log; }
    ;;
  *) echo "usage: tools/ab.sh soak|bench|tower|suite OUT [arm ...]"; exit 2 ;;
esac
