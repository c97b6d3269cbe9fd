"""Where does the ping-pong GEMM's time go?  Needs the ablation build (make -C lossyless_amd/csrc ablation).
LLA_GEMM_DEBUG=9: every wave stamps s_memtime after each barrier; wave 0 (upper wave row) and wave 4
(lower) of 8 workgroups report, per phase, the summed length of their load interval (partner computing)
and matrix interval (own MFMAs), and the time from the last matrix interval of a tile to the end of the
epilogue.  LLA_GEMM_DEBUG=1/2/3 time the kernel without LDS-DMA / MFMAs / fragment reads.
usage (GPU box): python tools/gemm_pp_trace.py [M=51200]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "lossyless_amd", "liblossyless_amd_ablation.so")
SHAPES = [("qkv", 2304, 768, 0), ("out", 768, 768, 2), ("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2)]


def child(mode):
    import torch
    buf = torch.zeros(8 * 2 * 32 + 4 * 512, dtype=torch.int64, device="cuda")
    os.environ["LLA_GEMM_TRACE"] = str(buf.data_ptr())
    sys.path.insert(0, ROOT)
    from lossyless_amd import _lib
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 51200
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, K, epi in SHAPES:
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda")
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
        run = lambda: L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 1 if mode in ("9", "11", "12", "14") else 5
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        line = f"[dbg {mode}] {name}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s"
        if mode in ("9", "11", "12", "14"):
            t = buf[:512].view(8, 2, 32).cpu().numpy().astype("float64")
            w = t[0]
            NI, its = int(w[0, 15]), w[0, 14]
            nk = K // 64
            tiles = its / nk
            print(line + f"   [s_memtime runs at {w[0,12]/max(w[0,13],1)*100:.0f} MHz; workgroup 0 active {w[0,13]/100:.1f} us]")
            for r, lab in ((0, "upper row"), (1, "lower row")):
                ld = [w[r, 2 * p] / its for p in range(NI)]
                mx = [w[r, 2 * p + 1] / its for p in range(NI)]
                print(f"    {lab}: barrier -> end of matrix segment, per phase " + " ".join(f"{x:5.0f}" for x in ld) +
                      " | end of matrix segment -> next barrier passed " + " ".join(f"{x:5.0f}" for x in mx) +
                      f" | middle K-tile {sum(ld)+sum(mx):6.0f} cyc (ideal {NI*512}) | epilogue + fragment re-read {w[r, 2*NI]/max(w[r,2*NI+1],1):7.0f} cyc x {w[r,2*NI+1]:.0f} tiles")
            print(f"    upper row, load segment after the LAST matrix segment of a K-tile: wait to start {w[0,19]/its:5.0f} | 3 reads issued {w[0,16]/its:5.0f}"
                  f" | 2 DMA + cursors {w[0,17]/its:5.0f} | lgkmcnt(0) {w[0,18]/its:5.0f}")
        elif mode == "8":
            t = buf[:512].view(8, 2, 16).cpu().numpy().astype("float64")
            print(line)
            for r, lab in ((0, "upper row"), (1, "lower row")):
                w = t[0, r]
                n = max(w[8], 1)
                print(f"    {lab}: load segment of phase 2 (middle K-tiles), cycles: 1 fragment read issued {w[0]/n:5.0f} | 2 DMA issued "
                      f"{w[1]/n:5.0f} | lgkmcnt(0) {w[2]/n:5.0f} | barrier {w[3]/n:5.0f} || matrix segment of phase 1: 8 MFMA + 3 reads issued "
                      f"{w[4]/n:5.0f} | barrier {w[5]/n:5.0f}")
        elif mode == "0":   # plain kernel: per-workgroup start / end stamps (100 MHz) of the LAST launch
            sp = buf[512:1536].view(512, 2).cpu().numpy().astype("float64")
            cy = buf[1536:].view(512, 2).cpu().numpy().astype("float64")
            cy = cy[sp[:, 1] > 0]
            sp = sp[sp[:, 1] > 0]
            t0 = sp[:, 0].min()
            st, en = (sp[:, 0] - t0) / 100, (sp[:, 1] - t0) / 100
            dur = en - st
            import numpy as np
            q = lambda a: " ".join(f"{np.percentile(a, x):6.1f}" for x in (0, 10, 50, 90, 100))
            print(line + f"   [{len(sp)} workgroups; us, percentiles 0/10/50/90/100: start {q(st)} | end {q(en)} | active {q(dur)}]")
            xcd = np.arange(len(sp)) % 8
            print("    end of the slowest workgroup per XCD: " + " ".join(f"{en[xcd == x].max():6.1f}" for x in range(8)) +
                  " | mean active per XCD: " + " ".join(f"{dur[xcd == x].mean():6.1f}" for x in range(8)) +
                  " | shader clock per XCD (s_memtime / s_memrealtime, MHz): " +
                  " ".join(f"{((cy[:, 1] - cy[:, 0]) / (sp[:, 1] - sp[:, 0]))[xcd == x].mean() * 100:5.0f}" for x in range(8)))
        else:
            print(line)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        if not os.path.exists(ABL):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "lossyless_amd", "csrc"), "ablation"])
        for mode in (sys.argv[2:] or ["0", "9", "1", "2", "4", "5"]):
            env = dict(os.environ, LLA_LIB=ABL, LLA_GEMM_DEBUG=mode)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", mode] + sys.argv[1:2], env=env)
# usage note: python tools/gemm_pp_trace.py [M] [modes...]
