"""A/B of the four-wave GEMM (gemm_q4.hip) against the ping-pong kernel at the tower's layer shapes: time per launch
and bitwise checksums of the outputs (every kernel accumulates K in the same order: they must be equal).
usage (GPU box): python tools/q4_probe.py [M=217600] [iters=20] [variants: pp q4:0 q4:1 q4:2 ...]
Each variant runs in its own interpreter (the switches are read once per process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("qkv f16", 2304, 768, 0), ("out resid", 768, 768, 2), ("fc1 gelu", 3072, 768, 1), ("fc2 resid", 768, 3072, 2)]


def child(M, iters):
    import torch
    from lossyless_amd import _lib
    L = _lib.lib()
    out = []
    for name, N, K, epi in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(N + K + epi)
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda")
        C0 = torch.randn(M, N, generator=g, device="cuda") if epi == 2 else None
        C = C0.clone() if epi == 2 else torch.zeros(M, N, dtype=torch.float16, device="cuda")
        st = _lib.stream_ptr()
        rc = L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        bits = C.view(torch.int16 if C.dtype == torch.float16 else torch.int32).long().flatten()
        w = torch.arange(bits.numel(), device="cuda") % 8191 + 1
        sums = [int(bits.sum()), int((bits * w).sum())]
        # spot check against fp64 on 64 rows
        rows = torch.arange(0, M, max(M // 64, 1), device="cuda")[:64]
        ref = A[rows].double() @ W.double().t() + bias.double()
        if epi == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        if epi == 2:
            ref = ref + C0[rows].double()
        err = float((C[rows].double() - ref).abs().max())
        del bits, w
        for _ in range(3):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out.append(dict(name=name, us=round(ms * 1e3, 1), tflops=round(2.0 * M * N * K / ms / 1e9, 1), sums=sums,
                        err=err))
        del A, W, C, C0
    print("RESULT " + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]), int(sys.argv[3]))
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 217600
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    variants = sys.argv[3:] or ["pp", "q4:0", "q4:1", "q4:2"]
    rounds = int(os.environ.get("Q4_PROBE_ROUNDS", "2"))
    res = {}
    for r in range(rounds):          # interleaved rounds: the box's clock state drifts
        for v in variants:
            env = dict(os.environ)
            if v.startswith("q4"):
                env["LLA_GEMM_Q4"] = "1"
                env["LLA_Q4_SCHED"] = v.split(":")[1] if ":" in v else "0"
                for kv in v.split(":")[2:]:
                    k, val = kv.split("=")
                    env[k] = val
            else:
                env["LLA_GEMM_Q4"] = "0"
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(M), str(iters)], env=env,
                               capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                print(f"{v}: FAILED rc={p.returncode}\n{p.stdout[-800:]}\n{p.stderr[-1500:]}")
                continue
            res.setdefault(v, []).append(json.loads(line[0][7:]))
    ref = res.get("pp", [None])[0]
    for v in variants:
        if v not in res:
            continue
        runs = res[v]
        tot = [sum(s["us"] for s in run) for run in runs]
        fl = sum(2.0 * M * N * K for _, N, K, _ in SHAPES)
        cells = []
        for i, (name, N, K, epi) in enumerate(SHAPES):
            best = min(run[i]["us"] for run in runs)
            same = "" if ref is None or v == "pp" else ("  ==pp" if runs[0][i]["sums"] == ref[i]["sums"] else "  !=pp")
            cells.append(f"{name} {best:7.1f} us {2.0 * M * N * K / best / 1e6:7.1f} TF err {runs[0][i]['err']:.1e}{same}")
        print(f"{v:>8}: layer {min(tot):8.1f} us = {fl / min(tot) / 1e6:7.1f} TFLOP/s | " + " | ".join(cells))


if __name__ == "__main__":
    main()
