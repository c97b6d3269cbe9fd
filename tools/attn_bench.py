"""Micro-benchmark of lla_attention50 / lla_layernorm768 at the tower's shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = _lib.lib()
qkv = torch.randn(B * 50, 2304, device="cuda").half()
o = torch.empty(B * 50, 768, dtype=torch.float16, device="cuda")
x = torch.randn(B * 50, 768, device="cuda")
w = torch.ones(768, device="cuda")
bb = torch.zeros(768, device="cuda")


def timeit(fn, it=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


st = _lib.stream_ptr()
t = timeit(lambda: L.lla_attention50(_lib.ptr(qkv), _lib.ptr(o), B, st))
print(f"attention50 B={B}: {t:.1f} us  {(qkv.numel() + o.numel()) * 2 / t / 1e6:.2f} TB/s")
t = timeit(lambda: L.lla_layernorm768(_lib.ptr(x), 768, _lib.ptr(w), _lib.ptr(bb), _lib.ptr(o), B * 50, st))
print(f"layernorm768 rows={B*50}: {t:.1f} us  {(x.numel() * 4 + o.numel() * 2) / t / 1e6:.2f} TB/s")
