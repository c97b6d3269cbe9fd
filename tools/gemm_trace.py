"""Where does a K-tile's time go?  LLA_GEMM_DEBUG=9 builds stamp s_memtime (100 MHz constant clock on
gfx950: 10 ns ticks) in wave 0 of 8 workgroups before the vmcnt wait, after the barrier and after the
last MFMA group of every K-tile.  usage (GPU box): python tools/gemm_trace.py [M=51200]"""
import os
import sys

import torch

buf = torch.zeros(8 * 128 * 4, dtype=torch.int64, device="cuda")
os.environ["LLA_GEMM_DEBUG"] = "9"
os.environ["LLA_GEMM_TRACE"] = str(buf.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in [("qkv", 2304, 768, 0), ("out", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]:
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
    for _ in range(3):
        L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    torch.cuda.synchronize()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    t = buf.view(8, 128, 4).cpu().numpy()
    nk = K // 64
    w = t[0]                                     # workgroup 0 (XCD 0, slot 0)
    n_it = int((w[:, 0] > 0).sum())
    w = w[:n_it].astype("float64")
    ckt = (t[0][:n_it, 3] & 255)
    vmwait = ((t[0][:n_it, 3] >> 8) & 0xffff).astype("float64")   # cycles in s_waitcnt vmcnt (own pieces)
    real = (t[0][:n_it, 3] >> 24).astype("float64")      # 100 MHz ticks
    mhz = (w[-1, 2] - w[0, 0]) / ((real[-1] - real[0]) / 100.0)   # shader cycles per microsecond
    ns = 1e3 / mhz                                # ns per shader cycle
    wait = w[:, 1] - w[:, 0]                      # vmcnt wait + barrier
    work = w[:, 2] - w[:, 1]                      # (epilogue on K-tile 0) + fetch + DMA issue + MFMA
    gap = w[1:, 0] - w[:-1, 2]                    # loop bookkeeping between iterations
    first = ckt == 0
    tot = wait.sum() + work.sum() + gap.sum()
    print(f"{name}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {n_it} K-tiles traced, nk={nk}, shader clock {mhz:.0f} MHz")
    print(f"   per K-tile cycles: wait+barrier {wait.mean():6.0f} (K-tile 0: {wait[first].mean():6.0f}, others {wait[~first].mean():6.0f})"
          f" | work {work.mean():6.0f} (K-tile 0 incl. epilogue: {work[first].mean():6.0f}, others {work[~first].mean():6.0f})"
          f" | between {gap.mean():5.0f}")
    print(f"   of the wait: own DMA pieces (s_waitcnt vmcnt) {vmwait.mean():6.0f} cycles, barrier {wait.mean() - vmwait.mean():6.0f}")
    print(f"   shares: wait {100*wait.sum()/tot:.1f} %, epilogue {100*(work[first].mean()-work[~first].mean())*first.sum()/tot:.1f} %,"
          f" MFMA-bound minimum 2 x {40 if K else 0} MFMA x 32 = {80*32} cycles vs work {work[~first].mean():.0f}; traced span {tot*ns/1e3:.1f} us")
