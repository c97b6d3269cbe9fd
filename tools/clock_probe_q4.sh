#!/bin/bash
# MFMA-busy fraction and shader clock: hipBLASLt vs the ping-pong kernel vs the four-wave kernel (and its timing
# ablations), the four layer shapes at M = 217 600, one process each.  usage: bash tools/clock_probe_q4.sh [tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$REPO/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"
run() { name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$name -o $name -- python $REPO/tools/gemm_bench.py 217600 12 > $OUT/$name.txt 2>&1; }
if [ -z "$SKIP_BASE" ]; then
timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/blas -o blas -- python $REPO/tools/blas_ceiling.py 217600 > $OUT/blas.txt 2>&1
run pp LLA_GEMM_Q4=0
run q4 LLA_GEMM_Q4=1 LLA_Q4_SCHED=${LLA_Q4_SCHED:-1}
run q4_serial_epilogue LLA_GEMM_Q4=1 LLA_Q4_PIPE=0   # (QKV / FC1: the fp16 epilogue between the tiles instead of in the K loop's shadows)
fi
for d in ${Q4_DBGS:-1 3 13}; do run q4_dbg$d LLA_GEMM_Q4=1 LLA_Q4_DBG=$d; done
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
for tag in sorted(os.listdir(out)):
    if not os.path.isdir(os.path.join(out, tag)):
        continue
    tr = glob.glob(f"{out}/{tag}/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(f"{out}/{tag}/**/*counter_collection.csv", recursive=True)
    if not tr or not cc:
        print(tag, "no output"); continue
    dur = {}
    for r in csv.DictReader(open(tr[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        d, name, grid = dur.get(r["Dispatch_Id"], (0, r["Kernel_Name"], "?"))
        if d < 150000:
            continue
        short = name.replace("lla::(anonymous namespace)::", "").replace("void ", "")[:44]
        agg[(short, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[(short, grid)]["dur_ns"].append(d)
    for key, c in agg.items():
        n = len(c["GRBM_GUI_ACTIVE"])
        if not n:
            continue
        m = lambda k: sum(c[k]) / max(len(c[k]), 1)
        dur_ns, grbm = m("dur_ns"), m("GRBM_GUI_ACTIVE")
        ghz = grbm / 8 / dur_ns
        wc = max(m("SQ_WAVE_CYCLES"), 1)
        print(f"{tag:10s} {key[0]:44s} grid {key[1]:>6s} n={n:3d} {dur_ns/1e3:8.1f} us  {ghz:.2f} GHz  MFMA busy {m('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (grbm / 8):.3f}"
              f"  wait_any {m('SQ_WAIT_ANY')/wc:.2f} wait_inst {m('SQ_WAIT_INST_ANY')/wc:.2f} active {m('SQ_ACTIVE_INST_ANY')/wc:.2f}")
PY
