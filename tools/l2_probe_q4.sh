#!/bin/bash
# L2 / fabric traffic of hipBLASLt vs the ping-pong kernel vs the four-wave kernel at M = 217 600 (separate PMC passes).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$REPO/gpurun_out/l2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; pmc=$2; shift; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/gemm_bench.py 217600 6 > $OUT/$name.txt 2>&1; }
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "hit:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  pn=${pass%%:*}; pc=${pass#*:}
  timeout 300 rocprofv3 --kernel-trace --pmc $pc --output-format csv -d $OUT/blas_$pn -o blas_$pn -- python $REPO/tools/blas_ceiling.py 217600 > $OUT/blas_$pn.txt 2>&1
  run pp_$pn "$pc" LLA_GEMM_Q4=0
  run q4_$pn "$pc" LLA_GEMM_Q4=1 LLA_Q4_SCHED=1
done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
rows = collections.defaultdict(dict)
for tag in sorted(os.listdir(out)):
    if not os.path.isdir(os.path.join(out, tag)):
        continue
    tr = glob.glob(f"{out}/{tag}/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True)
    if not tr or not cc:
        continue
    dur = {}
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        d = dur.get(r["Dispatch_Id"], 0)
        if d < 150000 or not ("gemm" in r["Kernel_Name"] or "Cijk" in r["Kernel_Name"]):
            continue
        name = r["Kernel_Name"].replace("lla::(anonymous namespace)::", "").replace("void ", "")[:28]
        # key by kernel + duration bucket (the library uses one kernel name for all four shapes)
        agg[(name, round(d / 100e3))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, c in sorted(agg.items()):
        print(tag, key[0], f"~{key[1]*100} us", {k: round(sum(v) / len(v) / (1024 if 'SIZE' in k else 1e6), 1) for k, v in c.items()}, "(SIZE: MB, counts: millions)")
PY
