#!/bin/bash
# Shader clock and MFMA-busy fraction of the experimental four-wave GEMM: full kernel and the no-operand-traffic ablation.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/clockq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LLA_LIB=$REPO/lossyless_amd/variants/liblossyless_amd_abl.so LLA_GEMM_QUAD=1 LLA_GEMM_EPILOGUE=direct
for d in 0 1; do
  LLA_QUAD_DBG=$d timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/d$d -o q -- python $REPO/tools/quad_check.py > $OUT/d$d.txt 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for tag in ("d0", "d1"):
    tr = glob.glob(f"gpurun_out/clockq/{tag}/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(f"gpurun_out/clockq/{tag}/**/*counter_collection.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(tr)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        d, name = dur.get(r["Dispatch_Id"], (0, r["Kernel_Name"]))
        if "gemm_quad" not in name:
            continue
        key = (name.split("gemm_quad_kernel")[1][:12], round(d / 2e5))   # group by epilogue and ~duration bucket
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["dur_ns"].append(d)
    for key, c in sorted(agg.items()):
        n = len(c["GRBM_GUI_ACTIVE"])
        if n < 5:
            continue
        dur_ns = sum(c["dur_ns"]) / len(c["dur_ns"])
        grbm = sum(c["GRBM_GUI_ACTIVE"]) / n
        mfma = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
        print(f"LLA_QUAD_DBG={tag[1]} gemm_quad_kernel{key[0]:12s} n={n:3d} dur {dur_ns/1e3:8.1f} us clock {grbm / 8 / dur_ns:.2f} GHz MFMA busy {mfma / 1024 / (grbm / 8):.3f}")
PY
