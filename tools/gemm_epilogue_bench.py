import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lossyless_amd import _lib
M = 217600
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    for epi, en in ((_lib.LLA_EPI_F16, "f16"), (_lib.LLA_EPI_QUICKGELU_F16, "gelu"), (_lib.LLA_EPI_RESID_F32, "resid")):
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
        st = _lib.stream_ptr()
        for _ in range(3):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name} {en:5s}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
