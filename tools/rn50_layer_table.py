"""Per-launch table of the RN50-CLIP tower: which convolution every kernel of one pass is, its algorithmic FLOPs and
activation bytes, and -- from a rocprofv3 kernel trace of `tools/rn50_bench.py B B 1` -- the rate each launch ran at.

usage:  python tools/rn50_layer_table.py [B=1024] [<kernel_trace.csv>]
        (GPU box)  rocprofv3 --kernel-trace -f csv -d gpurun_out/rn_trace -- python tools/rn50_bench.py 1024 1024 1
The launch order mirrors rn50_slices (csrc/rn50.hip) for the product switches (direct convolutions, fused downsample, one lane;
layer1's blocks as one kernel each when the trace holds bottleneck14_kernel launches);
the last complete pass of the trace is used."""
import csv
import glob
import os
import sys

BLOCKS, PLANES = (3, 4, 6, 3), (64, 128, 256, 512)


def launches(B, fused=True):
    """[(label, flops, bytes)] of one pass over B images, in launch order (fused: layer1's blocks as one kernel each)."""
    out = []

    def add(label, px, cin, cout, k=1, extra_read=0, in_px=None):
        macs = px * cin * cout * k * k
        rd = (in_px if in_px is not None else px) * cin * 2 + extra_read
        out.append((label, 2.0 * B * macs, float(B) * (rd + px * cout * 2)))

    add("stem.conv1 3x3 s2 (direct)", 112 * 112, 3, 32, 3, in_px=224 * 224)
    add("stem.conv2 3x3 (direct)", 112 * 112, 32, 32, 3)
    out.append(("stem.conv3 3x3 + avgpool (direct)", 2.0 * B * 112 * 112 * 32 * 64 * 9, float(B) * (112 * 112 * 32 * 2 + 56 * 56 * 64 * 2)))
    res, inpl = 56, 64
    for s, nb in enumerate(BLOCKS):
        p = PLANES[s]
        for b in range(nb):
            pre = f"layer{s + 1}.{b}."
            stride = 2 if (s > 0 and b == 0) else 1
            o = res // stride
            if fused and s == 0:
                px = res * res
                k3 = p + inpl if b == 0 else p       # block 0: conv3 | downsample over [t2 | x]
                out.append((pre + "bottleneck (fused)", 2.0 * B * px * (inpl * p + 9 * p * p + k3 * 4 * p), float(B) * px * (inpl + 4 * p) * 2))
                inpl = 4 * p
                continue
            add(pre + "conv1", res * res, inpl, p)
            add(pre + "conv2 3x3" + (" (direct)" if s == 0 else ""), res * res, p, p, 3)
            if b == 0:
                if s > 0:
                    out.append((pre + "avgpool main", 0.0, float(B) * (res * res * p * 2 + o * o * p * 2)))
                    out.append((pre + "avgpool input", 0.0, float(B) * (res * res * inpl * 2 + o * o * inpl * 2)))
                add(pre + "conv3 | downsample", o * o, p + inpl, 4 * p)
            else:
                add(pre + "conv3 + x", o * o, p, 4 * p, extra_read=o * o * 4 * p * 2)
            inpl, res = 4 * p, o
    E, T = 2048, 50
    out.append(("attnpool.tokens", 0.0, float(B) * (49 * E * 2 + T * E * 2)))
    out.append(("attnpool.kv", 2.0 * B * T * E * 2 * E, float(B) * (T * E * 2 + T * 2 * E * 2)))
    out.append(("attnpool.q", 2.0 * B * E * E, float(B) * (E * 2 * 2)))
    out.append(("attnpool.attend", 2.0 * B * 2 * T * E, float(B) * (T * 2 * E * 2 + 2 * E * 2)))
    out.append(("attnpool.c_proj", 2.0 * B * E * 1024, float(B) * (E * 2 + 1024 * 2)))
    return out


def trace_rows(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if name.startswith("at::") or "elementwise" in name or "Cijk" in name or "fill" in name.lower():
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    return rows


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rows = trace_rows(sys.argv[2]) if len(sys.argv) > 2 else None
    fused = rows is None or any("bottleneck14" in r[2] for r in rows)
    seq = launches(B, fused)
    if rows is not None:
        assert len(rows) % len(seq) == 0, (len(rows), len(seq))
        rows = rows[-len(seq):]
    tot_f = sum(f for _, f, _ in seq)
    tot_b = sum(b for _, _, b in seq)
    print(f"RN50-CLIP tower, {B} images, one pass: {len(seq)} launches, {tot_f / 1e12:.2f} TFLOP, {tot_b / 1e9:.1f} GB of activations")
    print(f"{'launch':38s} {'GFLOP':>8s} {'MB':>8s} {'flop/B':>7s}" + (f" {'us':>8s} {'TFLOP/s':>8s} {'TB/s':>6s}  kernel" if rows else ""))
    t_all = 0.0
    groups = {}
    for i, (label, fl, by) in enumerate(seq):
        line = f"{label:38s} {fl / 1e9:8.1f} {by / 1e6:8.1f} {fl / by:7.1f}"
        if rows:
            us = (rows[i][1] - rows[i][0]) / 1e3
            t_all += us
            line += f" {us:8.1f} {fl / us / 1e6:8.1f} {by / us / 1e6:6.2f}  {rows[i][2][:44]}"
            g = label.split(".")[0]
            a = groups.setdefault(g, [0.0, 0.0, 0.0])
            a[0] += us; a[1] += fl; a[2] += by
        print(line)
    if rows:
        print()
        for g, (us, fl, by) in groups.items():
            print(f"{g:10s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e6:5.2f} TB/s  {us / t_all * 100:5.1f} % of the kernel time")
        print(f"{'all':10s} {t_all:9.1f} us  {tot_f / t_all / 1e6:7.1f} TFLOP/s  {tot_b / t_all / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
