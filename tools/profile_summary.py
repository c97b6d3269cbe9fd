"""Condense the rocprofv3 outputs of tools/profile_round.sh into small text/JSON files
(kernel_stats.csv, pmc_by_kernel.csv, pmc_traffic.json) that are committed under profiles/."""
import collections
import csv
import glob
import gzip
import json
import os
import sys


def short(name):
    n = name.replace("lla::(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    if n.startswith("_ZN3lla12_GLOBAL__N_1"):
        import re
        m = re.match(r"_ZN3lla12_GLOBAL__N_1\d+([a-z0-9_]+)", n)
        n = m.group(1) if m else n
    return n[:70]


def main():
    root, tag = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3:   # a second kernel trace (e.g. the two-stream default command): kernel stats only
        return kernel_stats(root, sys.argv[3], f"kernel_stats_{sys.argv[3]}.csv")
    kernel_stats(root, "trace", "kernel_stats.csv")
    roofline_by_kernel(root, "trace", "roofline_by_kernel.csv")
    pmc_tables(root, tag)


def _traces(root, sub):
    """Kernel traces under root/sub: rocprofv3's `*kernel_trace.csv`, or the gzipped copy kept in profiles/."""
    return sorted(glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv"), recursive=True) +
                  glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv.gz"), recursive=True))


def _counters(root, sub):
    return sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True) +
                  glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv.gz"), recursive=True))


def _open(p):
    return gzip.open(p, "rt") if p.endswith(".gz") else open(p)


def kernel_stats(root, sub, out_name):
    # --- kernel stats from the kernel trace (own aggregation: exact avg/min/max per kernel)
    rows = collections.defaultdict(list)
    for p in _traces(root, sub):
        with _open(p) as f:
            for r in csv.DictReader(f):
                rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in rows.values()) or 1
    with open(os.path.join(root, out_name), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v) / 1e3, 1), round(sum(v) / len(v) / 1e3, 2),
                        round(min(v) / 1e3, 2), round(max(v) / 1e3, 2), round(100 * sum(v) / tot, 2)])


PEAK_TFLOPS, PEAK_TBS = 2500.0, 8.0      # dense fp16 MFMA, HBM3E (MI355X_MICROARCH.md)


def roofline_by_kernel(root, sub, out_name):
    """Per-kernel roofline fractions of the FULL-SIZE tower passes (8704 images = 435 200 rows), from the kernel trace:
    algorithmic FLOPs (GEMMs, against the dense fp16 MFMA peak) or algorithmic bytes (HBM-bound kernels, against 8 TB/s)
    divided by the mean duration of the full-size launches.  The trace does not carry GEMM shapes, so the layer GEMMs
    are told apart by their template arguments (epilogue) and, where out-proj and c_proj share an instantiation, by
    their order in the stream (they alternate); "full-size" = within -25 % / +50 % of the label's 90th-percentile launch."""
    rows = []
    for p in _traces(root, sub):
        with _open(p) as f:
            rows += list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    M = 8704 * 50
    w = 768
    spec = {   # label -> (bound, algorithmic work per full-size launch)
        "qkv (gemm_q4<EPI_F16>)": ("mfma", 2.0 * M * 3 * w * w),
        "c_fc + QuickGELU (gemm_q4<EPI_QGELU>)": ("mfma", 2.0 * M * 4 * w * w),
        "qkv (gemm_w8<EPI_F16>)": ("mfma", 2.0 * M * 3 * w * w),                      # round 6: the eight-wave kernel
        "c_fc + QuickGELU (gemm_w8<EPI_QGELU>)": ("mfma", 2.0 * M * 4 * w * w),
        "out-proj + residual [+ ln_2] (gemm_q4<EPI_RESID*>)": ("mfma", 2.0 * M * w * w),
        "c_proj + residual [+ ln_1] (gemm_q4<EPI_RESID*>)": ("mfma", 2.0 * M * 4 * w * w),
        "attention50_kernel": ("hbm", M * (3 * w + w) * 2.0),
        "layernorm768_kernel": ("hbm", None),     # rows from the launch's grid (4 rows per 256-thread workgroup): with LayerNorm in the
                                                  # GEMM epilogues only the class-token rows of the last block and ln_post are left
        "ln_pre_ln1_kernel": ("hbm", M * w * (4 + 4 + 2.0)),
        "patch embedding (gemm_pp<EPI_PATCH>)": ("mfma", 2.0 * 8704 * 49 * w * 3072),
    }
    dur = collections.defaultdict(list)
    ln_rows = []
    resid_turn = 0
    for r in rows:
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if k.startswith("gemm_q4_kernel<0,"): dur["qkv (gemm_q4<EPI_F16>)"].append(d)
        elif k.startswith("gemm_q4_kernel<1,"): dur["c_fc + QuickGELU (gemm_q4<EPI_QGELU>)"].append(d)
        elif k.startswith("gemm_w8_kernel<0,"): dur["qkv (gemm_w8<EPI_F16>)"].append(d)
        elif k.startswith("gemm_w8_kernel<1,"): dur["c_fc + QuickGELU (gemm_w8<EPI_QGELU>)"].append(d)
        elif k.startswith("gemm_q4_kernel<9,") or k.startswith("gemm_q4_kernel<2,"):
            dur[("out-proj" if resid_turn % 2 == 0 else "c_proj") + " + residual [+ ln_" + ("2" if resid_turn % 2 == 0 else "1") +
                "] (gemm_q4<EPI_RESID*>)"].append(d)
            resid_turn += 1
        elif k.startswith("gemm_pp_kernel<3,"):
            dur["patch embedding (gemm_pp<EPI_PATCH>)"].append(d)
            resid_turn = 0          # a pass starts here
        elif k in spec:
            dur[k].append(d)
            if k == "layernorm768_kernel": ln_rows.append((d, int(r["Grid_Size_X"]) // 256 * 4))
        elif k.startswith("lnx_cleanup_kernel"): dur["lnx_cleanup_kernel"].append(d)
    with open(os.path.join(root, out_name), "w") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "bound", "full_size_launches", "mean_us", "work_per_launch", "achieved", "unit", "frac_of_peak"])
        for label, v in dur.items():
            ref = sorted(v)[int(0.9 * (len(v) - 1))]      # (the 90th percentile, not the maximum: one slow outlier must not hide the rest)
            full = [d for d in v if 0.75 * ref <= d <= 1.5 * ref]
            mean = sum(full) / len(full)
            if label not in spec:
                wr.writerow([label, "-", len(full), round(mean / 1e3, 1), "", "", "", ""])
                continue
            bound, work = spec[label]
            if work is None:
                work = sum(n for d, n in ln_rows if 0.75 * ref <= d <= 1.5 * ref) / len(full) * w * (4 + 2.0)
            rate = work / (mean * 1e-9) / 1e12          # TFLOP/s or TB/s
            wr.writerow([label, bound, len(full), round(mean / 1e3, 1), f"{work:.4g}", round(rate, 2 if bound == "hbm" else 1),
                         "TB/s" if bound == "hbm" else "TFLOP/s", round(rate / (PEAK_TBS if bound == "hbm" else PEAK_TFLOPS), 4)])


def pmc_tables(root, tag):
    # --- PMC per kernel
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for p in _counters(root, "pmc_*"):
        with _open(p) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    with open(os.path.join(root, "pmc_by_kernel.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
        for k in sorted(agg):
            if not any(t in k for t in ("gemm", "attention", "layernorm", "ln_pre", "rans", "copy_streams")):
                continue
            for c in sorted(agg[k]):
                n = len(disp[k][c])
                w.writerow([k, c, n, round(agg[k][c] / n, 1)])
    # --- HBM traffic per GEMM launch (guide: bytes = SIZE * 1024; FETCH_SIZE reads half on gfx950), over the GEMM
    # launches of the FULL-SIZE tower passes only: the same launch mix as bench.py's `roofline` object (whose HIP events
    # time passes of `images_per_launch` images).  The run also holds 1024-image passes (calibration of the synthetic
    # affine, the verification batch); averaging over those as well -- as this script did until late in round 4 --
    # understates the per-launch figure by the ratio of the mean pass to the full one (8704 vs 4437 images in r04).
    # A pass starts at its patch-embedding GEMM (EPI_PATCH); `ln_pre_ln1_kernel`'s grid gives its rows (4 rows per
    # 256-thread workgroup), i.e. its images.
    def per_pass(sub, counter):
        rows = []
        for p in _counters(root, sub):
            with _open(p) as f:
                rows += [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        passes, cur = [], None
        for r in rows:
            k = short(r["Kernel_Name"])
            if k.startswith("gemm_pp_kernel<3,") or cur is None:
                cur = dict(images=0, gemm=[])
                passes.append(cur)
            if k.startswith("ln_pre_ln1_kernel"):
                cur["images"] = int(r["Grid_Size"]) // 256 * 4 // 50
            if "gemm" in k:
                cur["gemm"].append(float(r["Counter_Value"]))
        return [q for q in passes if q["images"]]
    pf, pw = per_pass("pmc_fetch", "FETCH_SIZE"), per_pass("pmc_write", "WRITE_SIZE")
    out = {}
    if pf and pw:
        full = max(q["images"] for q in pf)
        gf = [v for q in pf if q["images"] == full for v in q["gemm"]]
        gw = [v for q in pw if q["images"] == full for v in q["gemm"]]
        out = dict(gemm_bytes_per_launch=round(2 * sum(gf) * 1024 / len(gf) + sum(gw) * 1024 / len(gw)),
                   fetch_kb_raw_per_launch=round(sum(gf) / len(gf), 1), write_kb_per_launch=round(sum(gw) / len(gw), 1),
                   images_per_launch=full, passes=sum(q["images"] == full for q in pf), gemm_launches=len(gf),
                   note="FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide "
                        "coalesced read); WRITE_SIZE as reported; mean over the GEMM launches of the full-size tower "
                        "passes (`images_per_launch` images each): the launch mix of bench.py's roofline object",
                   tag=tag, kernel_source_sha=_source_sha())
    with open(os.path.join(root, "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(open(os.path.join(root, "kernel_stats.csv")).read())
    print(json.dumps(out))


def _source_sha():
    """bench.py's tag of the kernel sources (it refuses to print a traffic figure taken on other sources)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_sha()


if __name__ == "__main__":
    main()
