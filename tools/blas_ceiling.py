"""Library GEMM ceiling on this GPU for the tower's shapes (SURVEY.md 8d asks for a measured
hipBLASLt number next to the nominal MFMA peak).  torch.matmul on ROCm dispatches to
hipBLASLt / rocBLAS; plain GEMM only (no fused epilogue), fp16 in / fp16 out, fp32 accumulate."""
import sys

import torch


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
    g = torch.Generator(device="cuda").manual_seed(0)
    tot_ms = tot_fl = 0.0
    for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        for _ in range(5):
            C = A @ W.t()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30):
            C = A @ W.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        fl = 2.0 * M * N * K
        tot_ms += ms
        tot_fl += fl
        print(f"{name} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  (torch.matmul / hipBLASLt)")
    print(f"layer total: {tot_ms*1e3:8.1f} us  {tot_fl/tot_ms/1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
