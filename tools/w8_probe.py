"""A/B of the eight-wave GEMM on v_mfma_f32_16x16x32_f16 (gemm_w8.hip) against the product's four-wave kernel
(gemm_q4.hip) at the tower's fp16-output layer shapes (QKV, c_fc + QuickGELU): time per launch, error against fp64,
and the largest difference between the two kernels' outputs in fp16 ulps (the MFMA shapes round differently).

usage (GPU box; needs `make ablation`):  python tools/w8_probe.py [M=217600] [iters=20] [variants: q4 w8 w8:3 w8:13 ...]
  q4 = LLA_GEMM_W8=0 (the product path), w8 = LLA_GEMM_W8=1, w8:p0 = its serial epilogue, w8:<dbg> = the timing ablations of gemm_w8_kernel (WRONG
  results: 1 no LDS-DMA after the prologue, 2 no barrier, 3 no epilogue, 4 no counted waits, 13 = 1 + 3).
Each variant runs in its own interpreter, the variants interleaved over W8_PROBE_ROUNDS (3) rounds (the clock drifts)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [("qkv f16", 2304, 768, 0), ("fc1 gelu", 3072, 768, 1)]


def child(M, iters, dump):
    import torch
    from lossyless_amd import _lib
    L = _lib.lib()
    out = []
    for name, N, K, epi in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(N + K + epi)
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda")
        C = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        st = _lib.stream_ptr()
        rc = L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        rows = torch.arange(0, M, max(M // 257, 1), device="cuda")[:257]
        ref = A[rows].double() @ W.double().t() + bias.double()
        if epi == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        err = float((C[rows].double() - ref).abs().max())
        bits = C.view(torch.int16).long().flatten()
        w = torch.arange(bits.numel(), device="cuda") % 8191 + 1
        sums = [int(bits.sum()), int((bits * w).sum())]
        del bits, w
        if dump:
            torch.save(C[rows].cpu(), f"{dump}_{N}.pt")
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
        out.append(dict(name=name, us=round(ms * 1e3, 1), tflops=round(2.0 * M * N * K / ms / 1e9, 1), err=err, sums=sums))
        del A, W, C
    print("RESULT " + json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 217600
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    variants = sys.argv[3:] or ["q4", "w8", "w8:3", "w8:13"]
    rounds = int(os.environ.get("W8_PROBE_ROUNDS", "3"))
    lib = os.environ.get("LLA_LIB") or os.path.join(ROOT, "lossyless_amd", "liblossyless_amd_ablation.so")
    tmp = os.environ.get("TMPDIR", "/tmp")
    res = {}
    for r in range(rounds):
        for v in variants:
            env = dict(os.environ, LLA_LIB=lib, LLA_GEMM_W8="0" if v == "q4" else "1")
            head = v.split(":")[0]
            if "@" in head:     # w8@name: a `make w8variant NAME=name` build of gemm_w8.hip
                env["LLA_LIB"] = os.path.join(ROOT, "lossyless_amd", "variants", f"liblossyless_amd_{head.split('@')[1]}.so")
            if ":" in v:
                opt = v.split(":")[1]
                if opt == "p0":
                    env["LLA_W8_PIPE"] = "0"          # serial epilogue (same bits as the pipelined one)
                else:
                    env["LLA_W8_DBG"] = opt
            dump = os.path.join(tmp, "w8_probe_" + v.replace(":", "_")) if r == 0 and v in ("q4", "w8") else ""
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(M), str(iters), dump], env=env,
                               capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                print(f"{v}: FAILED rc={p.returncode}\n{p.stdout[-800:]}\n{p.stderr[-1500:]}")
                continue
            res.setdefault(v, []).append(json.loads(line[0][7:]))
    for v in variants:
        if v not in res:
            continue
        runs = res[v]
        cells = []
        for i, (name, N, K, epi) in enumerate(SHAPES):
            us = sorted(run[i]["us"] for run in runs)
            cells.append(f"{name} best {us[0]:7.1f} median {us[len(us) // 2]:7.1f} us = {2.0 * M * N * K / us[0] / 1e6:7.1f} TF, "
                         f"err {runs[0][i]['err']:.2e}, same sums every run: {len({tuple(r[i]['sums']) for r in runs}) == 1}")
        print(f"{v:>6}: " + " | ".join(cells))
    if "q4" in res and "w8" in res and os.path.exists(os.path.join(tmp, f"w8_probe_w8_{SHAPES[0][1]}.pt")):
        import torch
        for name, N, K, epi in SHAPES:
            a = torch.load(os.path.join(tmp, f"w8_probe_q4_{N}.pt")).float()
            b = torch.load(os.path.join(tmp, f"w8_probe_w8_{N}.pt")).float()
            d = (a - b).abs()
            ulp = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -14) * 2.0 ** -10
            print(f"{name}: q4 vs w8 on 257 rows: {int((d > 0).sum())} of {d.numel()} outputs differ, max {float((d / ulp).max()):.2f} fp16 ulp, "
                  f"max abs {float(d.max()):.3e}")
        for i, (name, N, K, epi) in enumerate(SHAPES):
            same = {v: res[v][0][i]["sums"] == res["q4"][0][i]["sums"] for v in res
                    if v.startswith("w8") and (":" not in v or v.endswith(":p0"))}
            print(f"{name}: checksums of the WHOLE output equal to q4's: {same}")
            q = min(r[i]["us"] for r in res["q4"])
            for v in res:
                if v != "q4":
                    w = min(r[i]["us"] for r in res[v])
                    print(f"{name}: {v} / q4 time = {w / q:.3f} ({(q / w - 1) * 100:+.1f} % throughput)")


if __name__ == "__main__":
    main()
