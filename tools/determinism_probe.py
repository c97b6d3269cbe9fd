"""Race screen: the same tower pass / the same GEMM launched many times must give the same bits every time.
    python tools/determinism_probe.py tower [passes]      # whole tower, batch 1024 (env: LLA_VIT_STREAMS, LLA_GEMM_PP, ...)
    python tools/determinism_probe.py gemm [launches]     # the four layer GEMMs at M = 51200, one at a time
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "tower"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(0)
if mode == "tower":
    import hubconf
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    from lossyless_amd.compressor import SyntheticImages
    x = SyntheticImages(4096).device_batch(0, 1024, "cuda")
    # reference = the first result that two consecutive passes agree on (the very first pass of a process may
    # itself be an outlier)
    z0 = comp.clip(x).clone()
    for _ in range(20):
        z1 = comp.clip(x).clone()
        if torch.equal(z0, z1):
            break
        z0 = z1
    bad = shown = 0
    for i in range(n):
        z = comp.clip(x)
        d = (z != z0).any(dim=1).nonzero().flatten().tolist()
        if d:
            bad += 1
            if shown < 6:
                shown += 1
                print("pass", i, "rows differing", d[:8], "max abs", float((z.float() - z0.float()).abs().max()), flush=True)
    print(f"tower: {bad} of {n} passes differ (env LLA_VIT_STREAMS={os.environ.get('LLA_VIT_STREAMS')} "
          f"LLA_GEMM_PP={os.environ.get('LLA_GEMM_PP')})", flush=True)
else:
    L = _lib.lib()
    M = 51200
    shapes = [("qkv", 2304, 768, _lib.LLA_EPI_F16), ("out", 768, 768, _lib.LLA_EPI_RESID_F32),
              ("fc1", 3072, 768, _lib.LLA_EPI_QUICKGELU_F16), ("fc2", 768, 3072, _lib.LLA_EPI_RESID_F32)]
    for name, N, K, epi in shapes:
        A = (torch.randn(M, K, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, device="cuda") * 0.05).half()
        bias = torch.randn(N, device="cuda")
        f32 = epi == _lib.LLA_EPI_RESID_F32
        base = torch.randn(M, N, device="cuda") if f32 else None

        def run():
            C = base.clone() if f32 else torch.empty(M, N, dtype=torch.float16, device="cuda")
            _lib.check(L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi,
                                      _lib.stream_ptr("cuda")), "gemm")
            return C
        c0 = run()
        bad = 0
        for i in range(n):
            c = run()
            if not torch.equal(c, c0):
                bad += 1
                idx = (c != c0).nonzero()
                rows = sorted(set(idx[:, 0].tolist()))
                cols = sorted(set(idx[:, 1].tolist()))
                print(name, "launch", i, "elements", len(idx), "rows", rows[:6], "..", rows[-1], "cols", cols[:4], "..", cols[-1], flush=True)
        print(f"gemm {name}: {bad} of {n} launches differ", flush=True)

    # attention and LayerNorm at the tower's sizes
    B = 1024
    qkv = (torch.randn(B * 50, 2304, device="cuda") * 0.7).half()
    x = torch.randn(B * 50, 768, device="cuda")
    w, b = torch.randn(768, device="cuda"), torch.randn(768, device="cuda")

    def attn():
        o = torch.empty(B * 50, 768, dtype=torch.float16, device="cuda")
        _lib.check(L.lla_attention50(_lib.ptr(qkv), _lib.ptr(o), B, _lib.stream_ptr("cuda")), "attn")
        return o

    def ln():
        y = torch.empty(B * 50, 768, dtype=torch.float16, device="cuda")
        _lib.check(L.lla_layernorm768(_lib.ptr(x), 768, _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), B * 50,
                                      _lib.stream_ptr("cuda")), "ln")
        return y
    for name, fn in (("attention", attn), ("layernorm", ln)):
        r0 = fn()
        bad = sum(0 if torch.equal(fn(), r0) else 1 for _ in range(n))
        print(f"{name}: {bad} of {n} launches differ", flush=True)
