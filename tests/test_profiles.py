"""profiles/ is evidence only if it can be recomputed: the per-kernel tables of a round (`rNN_kernel_stats.csv`,
`rNN_roofline_by_kernel.csv`) must come out of the TRACKED rocprofv3 kernel trace (`profiles/rNN_trace/`, gzipped) through
tools/profile_summary.py byte for byte, and the bench line's `roofline` must agree with the trace's GEMM durations.
Rounds 5 and 6."""
import csv
import filecmp
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PROFILES = os.path.join(ROOT, "profiles")


def _summary():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_summary
    return profile_summary


def test_round5_kernel_tables_are_reproduced_from_the_tracked_trace(tmp_path):
    ps = _summary()
    root = str(tmp_path / "r05")
    shutil.copytree(os.path.join(PROFILES, "r05_trace"), root)
    ps.kernel_stats(root, "trace", "kernel_stats.csv")
    ps.roofline_by_kernel(root, "trace", "roofline_by_kernel.csv")
    assert filecmp.cmp(os.path.join(root, "kernel_stats.csv"), os.path.join(PROFILES, "r05_kernel_stats.csv"), shallow=False)
    assert filecmp.cmp(os.path.join(root, "roofline_by_kernel.csv"), os.path.join(PROFILES, "r05_roofline_by_kernel.csv"), shallow=False)
    # the PMC passes behind `roofline.traffic` (FETCH_SIZE / WRITE_SIZE, one pass each): same figure, same launch mix
    ps.pmc_tables(root, "r05")
    with open(os.path.join(root, "pmc_traffic.json")) as f:
        got = json.load(f)
    with open(os.path.join(PROFILES, "r05_pmc_traffic.json")) as f:
        want = json.load(f)
    for k in ("gemm_bytes_per_launch", "fetch_kb_raw_per_launch", "write_kb_per_launch", "images_per_launch", "passes",
              "gemm_launches"):
        assert got[k] == want[k], k


def test_round5_bench_line_agrees_with_the_table():
    with open(os.path.join(PROFILES, "r05_bench_1gpu.json")) as f:
        line = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
    roof = line["roofline"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["bound"] == "mfma"
    assert roof["gemm_ms_per_step"] <= line["ms_per_step"]
    with open(os.path.join(PROFILES, "r05_roofline_by_kernel.csv")) as f:
        rows = list(csv.DictReader(f))
    fracs = [float(r["frac_of_peak"]) for r in rows if r["bound"] == "mfma" and "gemm_q4" in r["kernel"]]
    # the line's class-wide fraction (live HIP events, no tracer) lies inside the per-kernel range of the traced run
    assert fracs and min(fracs) - 0.02 <= roof["frac"] <= max(fracs) + 0.02
    assert line["config"]["verified"] is True


def test_round6_kernel_tables_are_reproduced_from_the_tracked_trace(tmp_path):
    ps = _summary()
    root = str(tmp_path / "r06")
    shutil.copytree(os.path.join(PROFILES, "r06_trace"), root)
    ps.kernel_stats(root, "trace", "kernel_stats.csv")
    ps.roofline_by_kernel(root, "trace", "roofline_by_kernel.csv")
    assert filecmp.cmp(os.path.join(root, "kernel_stats.csv"), os.path.join(PROFILES, "r06_kernel_stats.csv"), shallow=False)
    assert filecmp.cmp(os.path.join(root, "roofline_by_kernel.csv"), os.path.join(PROFILES, "r06_roofline_by_kernel.csv"), shallow=False)
    ps.pmc_tables(root, "r06")
    with open(os.path.join(root, "pmc_traffic.json")) as f:
        got = json.load(f)
    with open(os.path.join(PROFILES, "r06_pmc_traffic.json")) as f:
        want = json.load(f)
    for k in ("gemm_bytes_per_launch", "fetch_kb_raw_per_launch", "write_kb_per_launch", "images_per_launch", "passes",
              "gemm_launches"):
        assert got[k] == want[k], k
    # round 6: QKV and c_fc run on the eight-wave kernel, the residual layers on the four-wave kernel
    with open(os.path.join(PROFILES, "r06_roofline_by_kernel.csv")) as f:
        kernels = [r["kernel"] for r in csv.DictReader(f)]
    assert "qkv (gemm_w8<EPI_F16>)" in kernels and "c_fc + QuickGELU (gemm_w8<EPI_QGELU>)" in kernels
    assert not any("gemm_q4<EPI_F16>" in k or "gemm_q4<EPI_QGELU>" in k for k in kernels)


def test_round6_bench_line_agrees_with_the_table():
    with open(os.path.join(PROFILES, "r06_bench_1gpu.json")) as f:
        line = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
    roof = line["roofline"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["bound"] == "mfma"
    assert roof["gemm_ms_per_step"] <= line["ms_per_step"]
    with open(os.path.join(PROFILES, "r06_roofline_by_kernel.csv")) as f:
        rows = list(csv.DictReader(f))
    fracs = [float(r["frac_of_peak"]) for r in rows if r["bound"] == "mfma" and ("gemm_q4" in r["kernel"] or "gemm_w8" in r["kernel"])]
    assert fracs and min(fracs) - 0.02 <= roof["frac"] <= max(fracs) + 0.02
    assert line["config"]["verified"] is True and line["verification"]["oracle_pin"].startswith("unpinned")
    # `traffic` is printed only for the kernel sources the PMC passes were taken on (lla_source_sha)
    with open(os.path.join(PROFILES, "r06_pmc_traffic.json")) as f:
        pmc = json.load(f)
    if roof["traffic"] is not None:
        assert roof["traffic"] == pmc["gemm_bytes_per_launch"]
