#!/bin/bash
# MFMA-busy fraction and shader clock of the library GEMM vs this library's, same shapes (M = 217 600), one process each.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/blas -o blas -- python $REPO/tools/blas_ceiling.py 217600 > $OUT/blas.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/ours -o ours -- python $REPO/tools/gemm_bench.py 217600 20 > $OUT/ours.txt 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
for tag in ("blas", "ours"):
    tr = glob.glob(f"gpurun_out/clock/{tag}/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(f"gpurun_out/clock/{tag}/**/*counter_collection.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(tr)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        d, name, grid = dur.get(r["Dispatch_Id"], (0, r["Kernel_Name"], "?"))
        if d < 150000:
            continue
        key = (name[:60], grid)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["dur_ns"].append(d)
    for key, c in agg.items():
        n = len(c["GRBM_GUI_ACTIVE"])
        if not n:
            continue
        dur_ns = sum(c["dur_ns"]) / len(c["dur_ns"])
        grbm = sum(c["GRBM_GUI_ACTIVE"]) / n
        mfma = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
        ghz = grbm / 8 / dur_ns
        print(f"{tag} {key[0]:60s} grid {key[1]:>8s} n={n:3d} dur {dur_ns/1e3:8.1f} us clock {ghz:.2f} GHz MFMA busy {mfma / 1024 / (grbm / 8):.3f}")
PY
