"""Do the two workgroups that share a CU under gemm_duo_kernel run their epilogues at the same time?
Needs the ablation build.  usage (GPU box): python tools/duo_trace.py [M=51200]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "lossyless_amd", "liblossyless_amd_ablation.so")


def child():
    import numpy as np
    import torch
    buf = torch.zeros(4096 + 512 * 40, dtype=torch.int64, device="cuda")
    os.environ["LLA_GEMM_TRACE"] = str(buf.data_ptr())
    sys.path.insert(0, ROOT)
    from lossyless_amd import _lib
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 51200
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, K, epi in [("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1)]:
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda")
        C = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        run = lambda: L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        buf.zero_()
        run()
        torch.cuda.synchronize()
        t = buf[4096:].view(512, 40).cpu().numpy()
        t = t[t[:, 1] > 0]
        hw, xcc = t[:, 0] & 0xFFFFFFFF, t[:, 0] >> 32
        cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)   # XCC, SE, SH, CU
        t0 = t[:, 1].min()
        print(f"{name}: {len(t)} workgroups on {len(set(cu.tolist()))} distinct (XCC,SE,SH,CU); "
              f"wave slots seen {sorted(set((hw & 15).tolist()))}; kernel {(t[:, 3].max() - t0) / 100:.1f} us")
        both, alone, gaps = 0.0, 0.0, []
        for c in sorted(set(cu.tolist())):
            w = t[cu == c]
            if len(w) != 2:
                continue
            iv = []
            for r in w:
                n = int(min(r[2], 16))
                iv.append([(r[4 + 2 * j] - t0, r[5 + 2 * j] - t0) for j in range(n)])
            a, b = iv
            ov = sum(max(0, min(x[1], y[1]) - max(x[0], y[0])) for x in a for y in b)
            tot = sum(x[1] - x[0] for x in a) + sum(y[1] - y[0] for y in b)
            both += 2 * ov
            alone += tot - 2 * ov
            gaps.append(abs(a[0][0] - b[0][0]) / 100)
        print(f"   epilogue time spent while the CU's other workgroup is ALSO in its epilogue: {both / max(both + alone, 1):.2f} of all "
              f"epilogue time; |start of first epilogue A - B| median {np.median(gaps):.2f} us, max {np.max(gaps):.2f} us; "
              f"mean epilogue {(both + alone) / 100 / max(int(t[:, 2].sum()), 1):.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for extra in ({}, {"LLA_GEMM_DUO_STAGGER": "1"}):
            env = dict(os.environ, LLA_LIB=ABL, LLA_GEMM_DUO="1", **extra)
            print("env", extra)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:2], env=env)
