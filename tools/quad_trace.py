"""Where does a K-tile of the four-wave GEMM go?  LLA_QUAD_DBG=9 (10: without operand traffic) stamps s_memtime in wave 0 of
8 workgroups at the top of a K-tile, after the wait + barrier, after k-step 0's MFMAs have been issued and at the end.
usage (GPU box, ablation build via LLA_LIB): LLA_QUAD_DBG=9 python tools/quad_trace.py"""
import os
import sys

import torch

buf = torch.zeros(8 * 128 * 4, dtype=torch.int64, device="cuda")
os.environ["LLA_GEMM_QUAD"] = "1"
os.environ["LLA_GEMM_EPILOGUE"] = "direct"
os.environ["LLA_GEMM_TRACE"] = str(buf.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402

M = 217600
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in [("qkv", 2304, 768, 0), ("fc2", 768, 3072, 0)]:
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    C = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    for _ in range(3):
        L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    t = buf.view(8, 128, 4).cpu().numpy().astype("float64")
    nk = K // 64
    for wg in (0, 3):
        w = t[wg]
        n_it = int((w[:, 0] > 0).sum())
        w = w[:n_it]
        ckt = (torch.arange(n_it) % nk).numpy()
        wait = w[:, 1] - w[:, 0]
        step0 = w[:, 2] - w[:, 1]
        rest = w[:, 3] - w[:, 2]
        gap = w[1:, 0] - w[:-1, 3]
        first = ckt == 0
        print(f"{name} wg {wg}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {n_it} K-tiles, nk={nk}; cycles per K-tile (ideal 512 + 1536): "
              f"wait+barrier {wait.mean():6.0f} | k-step 0 (fragment fetch + 16 MFMAs{' + DMA issue' if os.environ.get('LLA_QUAD_DBG') == '9' else ''}) "
              f"{step0[~first].mean():6.0f} (first K-tile of a tile, incl. epilogue: {step0[first].mean():6.0f}) | k-steps 1-3 {rest.mean():6.0f} | "
              f"between {gap.mean():5.0f} | total {(w[-1, 3] - w[0, 0]) / n_it:6.0f}")
