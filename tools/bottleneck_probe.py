"""The fused layer1 bottleneck (csrc/bottleneck_fused.hip) alone: time per launch at the tower's shape, against the three
kernels it replaces; with a -DLLA_BN_DBG=9 build (make -C lossyless_amd/csrc tuvariant TU=bottleneck_fused NAME=bn9 DEFS=-DLLA_BN_DBG=9,
LLA_LIB=lossyless_amd/variants/liblossyless_amd_bn9.so BN_TRACE=1) the shader-clock breakdown by phase.
usage (GPU box): python tools/bottleneck_probe.py [n=1024] [iters=10]

Per CU or the whole chip (profiles/r06_rn50_fused_bottleneck.txt section 3): the same trace from a build whose launcher starts G
workgroups instead of one per CU -- compile a copy of the file with `int grid = bn_cu_count();` replaced by `int grid = G;`
(-DLLA_BN_DBG=9, linked as `make tuvariant` links) and run with BN_GRID=G."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    H = W = 56
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(n, H, W, 256, generator=g, device="cuda").abs() * 0.6).half()
    w1 = torch.zeros(128, 256, dtype=torch.float16, device="cuda"); w1[:64] = (torch.randn(64, 256, generator=g, device="cuda") * 0.09).half()
    w2 = torch.zeros(128, 576, dtype=torch.float16, device="cuda"); w2[:64] = (torch.randn(64, 576, generator=g, device="cuda") * 0.06).half()
    w3 = (torch.randn(256, 64, generator=g, device="cuda") * 0.09).half()
    b1, b2, b3 = (torch.randn(k, generator=g, device="cuda") * 0.2 for k in (128, 128, 256))
    out = torch.empty_like(x)
    t1 = torch.empty(n * H * W, 64, dtype=torch.float16, device="cuda")
    t2 = torch.empty_like(t1)
    L, st = _lib.lib(), _lib.stream_ptr()
    M = n * H * W

    def fused():
        assert L.lla_rn50_bottleneck_f16(_lib.ptr(x), n, H, W, 256, 256, _lib.ptr(w1), 256, _lib.ptr(b1), _lib.ptr(w2), 576, _lib.ptr(b2),
                                         _lib.ptr(w3), 64, _lib.ptr(b3), _lib.ptr(out), 256, st) == 0

    def three():
        assert L.lla_gemm_f16_ex(_lib.ptr(x), 256, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(t1), 64, None, 0, M, 128, 256, _lib.LLA_EPI_RELU_F16, st) == 0
        assert L.lla_conv3x3_direct_relu_f16(_lib.ptr(t1), n, H, W, 64, 64, _lib.ptr(w2), 576, _lib.ptr(b2), _lib.ptr(t2), 64, 64, 0, st) == 0
        assert L.lla_gemm_f16_ex(_lib.ptr(t2), 64, _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(out), 256, _lib.ptr(x), 256, M, 256, 64,
                                 _lib.LLA_EPI_ADD_RELU_F16, st) == 0

    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    flops = 2.0 * M * (256 * 64 + 576 * 64 + 64 * 256)
    by = M * 512 * 2.0
    for name, fn in (("fused", fused), ("three kernels", three), ("fused", fused)):
        us = timed(fn)
        print(f"{name:14s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  {by / us / 1e6:5.2f} TB/s (x in + out)")
    if os.environ.get("BN_TRACE") == "1":
        fused()
        torch.cuda.synchronize()
        grid = int(os.environ.get("BN_GRID", "256"))         # (a build whose launcher starts fewer workgroups: per-CU rate without the other CUs)
        tr = out.view(torch.int64).flatten()[: grid * 4 * 8].reshape(grid, 4, 8).double().cpu()
        names = ["top wait + barrier", "conv1 loop", "conv1 epilogue + barrier", "identity loads + DMA issue", "conv2", "barrier, t2 write, barrier",
                 "conv3 + stores", "-"]
        tiles = n * 16 / grid
        tot = tr[:, :, :7].sum(-1).mean()
        print(f"shader clocks per tile (mean over workgroups, {tiles:.0f} tiles each; 100 MHz counter -> x 1e-2 us): total {tot / tiles:.0f}")
        for w in range(4):
            print(f"  wave {w}: " + ", ".join(f"{names[k]} {tr[:, w, k].mean() / tiles:.0f}" for k in range(7)))


if __name__ == "__main__":
    main()
