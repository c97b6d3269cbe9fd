"""Quad GEMM (LLA_GEMM_QUAD=1) vs the shipped selection: run once per setting, compare the printed checksums
(bit patterns of every output) and the timings.  usage: python tools/quad_check.py [M=217600]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 217600
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    for epi, en in ((_lib.LLA_EPI_F16, "f16"), (_lib.LLA_EPI_QUICKGELU_F16, "gelu"), (_lib.LLA_EPI_RESID_F32, "resid")):
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
        st = _lib.stream_ptr()
        L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        torch.cuda.synchronize()
        bits = C.view(torch.int32 if epi == 2 else torch.int16).to(torch.int64)
        chk = int(bits.sum()) ^ int((bits * torch.arange(1, N + 1, device="cuda")).sum())
        ref = (A[:512].float() @ W.float().t() + bias)
        if epi == 1:
            ref = ref * torch.sigmoid(1.702 * ref)
        err = float((C[:512].float() - ref).abs().max())
        for _ in range(2):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name} {en:5s} chk {chk & 0xffffffffffff:012x} err512 {err:.3e} | {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
