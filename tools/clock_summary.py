"""Shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA-busy fraction and wave-state fractions per kernel from the
rocprofv3 --kernel-trace --pmc outputs under a directory (one sub-directory per run): tools/clock_probe_q4.sh,
tools/clock_probe_w8.sh.  usage: python tools/clock_summary.py DIR"""
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
