set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/profiles_r03; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 17 --warmup 0 --no-verify --no-cpu-baseline --no-extra --no-profile"
for c in FETCH_SIZE WRITE_SIZE; do n=pmc_$(echo $c | tr A-Z a-z | sed 's/_size//'); rm -rf $OUT/$n; timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -o $n -- $BENCH > $OUT/$n.json 2> $OUT/$n.err; done
cd $REPO; python tools/profile_summary.py $OUT r03 > $OUT/summary.txt 2>&1; cat $OUT/pmc_traffic.json
python bench.py > gpurun_out/b6.json 2> gpurun_out/b6.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b6.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','verified')}); print(d['roofline'])
print(d['reference_call_stage'].get('gpu_preprocess_default_arguments')); print(d['stl10_shaped_stage'])
PY
