#!/usr/bin/env python3
"""Copy what `bash tools/profile_round.sh <tag>` left under gpurun_out/profiles_<tag>/ (scratch) into profiles/ (tracked):
the summaries as `<tag>_*`, `pmc_traffic.json` (what bench.py's `roofline.traffic` reads, with the kernel sources' sha), and
the raw kernel trace + FETCH_SIZE / WRITE_SIZE counter passes gzipped under profiles/<tag>_trace/ (tests/test_profiles.py
recomputes the tables from them).

  python tools/collect_profiles.py r05 [bench_line.json]
"""
import gzip
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    tag = sys.argv[1]
    src, dst = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}"), os.path.join(ROOT, "profiles")
    names = {"kernel_stats.csv": f"{tag}_kernel_stats.csv", "roofline_by_kernel.csv": f"{tag}_roofline_by_kernel.csv",
             "pmc_by_kernel.csv": f"{tag}_pmc_by_kernel.csv", "pmc_traffic.json": f"{tag}_pmc_traffic.json",
             "bench_under_trace.json": f"{tag}_bench_under_trace.json", "kernel_stats_rn50.csv": f"{tag}_rn50_kernel_stats.csv",
             "hipblaslt_ceiling.txt": f"{tag}_hipblaslt_ceiling.txt", "gemm_bench.txt": f"{tag}_gemm_bench.txt",
             "rn50_bench.txt": f"{tag}_rn50_bench.txt",
             os.path.join("trace", f"{tag}_kernel_stats.csv"): f"{tag}_rocprofv3_kernel_stats_raw.csv"}
    for a, b in names.items():
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
        else:
            print("missing:", a)
    shutil.copy(os.path.join(src, "pmc_traffic.json"), os.path.join(dst, "pmc_traffic.json"))
    for sub, name in (("trace", f"{tag}_kernel_trace.csv"), ("pmc_fetch", "pmc_fetch_counter_collection.csv"),
                      ("pmc_write", "pmc_write_counter_collection.csv")):
        os.makedirs(os.path.join(dst, f"{tag}_trace", sub), exist_ok=True)
        with open(os.path.join(src, sub, name), "rb") as f, \
                gzip.GzipFile(os.path.join(dst, f"{tag}_trace", sub, name + ".gz"), "wb", 9, mtime=0) as g:
            g.write(f.read())
    if len(sys.argv) > 2:
        shutil.copy(sys.argv[2], os.path.join(dst, f"{tag}_bench_1gpu.json"))


if __name__ == "__main__":
    main()
