"""Cycle trace of the four-wave GEMM (probe build, LLA_Q4_DBG=20): s_memtime stamps of wave 0 of workgroups 0..7 at
every K-tile start (1 first / 2 middle / 3 last), epilogue start (4) and epilogue end (5).
usage (GPU box): python tools/q4_trace.py [M=217600]"""
import os
import sys

import torch

buf = torch.zeros(8 * 1024, dtype=torch.int64, device="cuda")
os.environ["LLA_GEMM_Q4"] = "1"
os.environ["LLA_Q4_DBG"] = "20"
os.environ["LLA_Q4_TRACE"] = hex(buf.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 217600
L = _lib.lib()
for name, N, K, epi in [("qkv f16", 2304, 768, 0), ("out resid", 768, 768, 2), ("fc1 gelu", 3072, 768, 1), ("fc2 resid", 768, 3072, 2)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
    for _ in range(3):
        L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    buf.zero_()
    L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    torch.cuda.synchronize()
    t = buf.cpu().view(8, 512, 2)
    for wg in (0, 3):
        ev = [(int(a), int(b)) for a, b in t[wg] if int(b) != 0]
        d = {1: [], 2: [], 3: [], 4: [], 5: []}
        for (t0, tag0), (t1, tag1) in zip(ev, ev[1:]):
            d[tag0].append(t1 - t0)
        med = lambda v: sorted(v)[len(v) // 2] if v else 0
        print(f"{name} wg{wg}: events {len(ev)}  first K-tile {med(d[1])}  middle {med(d[2])} (min {min(d[2]) if d[2] else 0} max {max(d[2]) if d[2] else 0})  "
              f"last {med(d[3])}  epilogue {med(d[4])}  post-epilogue preload {med(d[5])} cycles; tiles {len(d[4])}")
