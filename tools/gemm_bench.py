"""Micro-benchmark of lla_gemm_f16 at the tower's shapes (run on the GPU box).
usage: python tools/gemm_bench.py [rows_per_chunk=12800] [iters=30]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    L = _lib.lib()
    shapes = [("qkv   f16  ", 2304, 768, _lib.LLA_EPI_F16),
              ("out   resid", 768, 768, _lib.LLA_EPI_RESID_F32),
              ("fc1   gelu ", 3072, 768, _lib.LLA_EPI_QUICKGELU_F16),
              ("fc2   resid", 768, 3072, _lib.LLA_EPI_RESID_F32)]
    g = torch.Generator(device="cuda").manual_seed(0)
    tot_ms, tot_fl = 0.0, 0.0
    for name, N, K, epi in shapes:
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda")
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
        st = _lib.stream_ptr()
        for _ in range(5):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            rc = L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        ms = e0.elapsed_time(e1) / iters
        fl = 2.0 * M * N * K
        tot_ms += ms
        tot_fl += fl
        print(f"{name} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
    print(f"layer total: {tot_ms*1e3:8.1f} us  {tot_fl/tot_ms/1e9:7.1f} TFLOP/s "
          f"(GLDS={os.environ.get('LLA_GEMM_GLDS','1')})")


if __name__ == "__main__":
    main()
