"""Debug probe: does kernel V (victim) produce different bits while kernel P (partner) runs on another stream?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.manual_seed(0)
M = 25600
dev = "cuda"
A768 = (torch.randn(M, 768, device=dev) * 0.5).half()
A3072 = (torch.randn(M, 3072, device=dev) * 0.5).half()
W = {k: (torch.randn(nn, kk, device=dev) * 0.05).half() for k, (nn, kk) in
     dict(qkv=(2304, 768), out=(768, 768), fc1=(3072, 768), fc2=(768, 3072)).items()}
bias = torch.randn(3072, device=dev)
x32 = torch.randn(M, 768, device=dev)
lnw, lnb = torch.randn(768, device=dev), torch.randn(768, device=dev)
qkv = (torch.randn(M, 2304, device=dev) * 0.7).half()
Cp = {k: torch.empty(M, 3072, dtype=torch.float16, device=dev) for k in ("a", "b")}
Xp = torch.randn(M, 768, device=dev)


def gemm(name, A, C, epi, s):
    N, K = W[name].shape
    _lib.check(L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W[name]), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, s), name)


def k_ln(out, s):
    _lib.check(L.lla_layernorm768(_lib.ptr(x32), 768, _lib.ptr(lnw), _lib.ptr(lnb), _lib.ptr(out), M, s), "ln")


def k_attn(out, s):
    _lib.check(L.lla_attention50(_lib.ptr(qkv), _lib.ptr(out), M // 50, s), "attn")


victims = {
    "layernorm": (k_ln, lambda: torch.empty(M, 768, dtype=torch.float16, device=dev)),
    "attention": (k_attn, lambda: torch.empty(M, 768, dtype=torch.float16, device=dev)),
    "gemm_qkv": (lambda o, s: gemm("qkv", A768, o, _lib.LLA_EPI_F16, s), lambda: torch.empty(M, 2304, dtype=torch.float16, device=dev)),
    "gemm_fc1": (lambda o, s: gemm("fc1", A768, o, _lib.LLA_EPI_QUICKGELU_F16, s), lambda: torch.empty(M, 3072, dtype=torch.float16, device=dev)),
}
partners = {
    "none": None,
    "gemm_qkv": lambda s: gemm("qkv", A768, Cp["a"], _lib.LLA_EPI_F16, s),
    "gemm_fc1": lambda s: gemm("fc1", A768, Cp["a"], _lib.LLA_EPI_QUICKGELU_F16, s),
    "gemm_fc2_resid": lambda s: gemm("fc2", A3072, Xp, _lib.LLA_EPI_RESID_F32, s),
    "gemm_out_resid": lambda s: gemm("out", A768, Xp, _lib.LLA_EPI_RESID_F32, s),
    "attention": lambda s: k_attn(Cp["b"], s),
    "layernorm": lambda s: k_ln(Cp["b"], s),
    "memcpy": lambda s: Cp["b"].copy_(Cp["a"], non_blocking=True),
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
which = sys.argv[2].split(",") if len(sys.argv) > 2 else list(victims)
for vname in which:
    vk, mk = victims[vname]
    ref = mk()
    vk(ref, _lib.stream_ptr(dev))
    torch.cuda.synchronize()
    for pname, pk in partners.items():
        bad = 0
        out = mk()
        for i in range(n):
            with torch.cuda.stream(sb):
                if pk is not None:
                    if pname == "memcpy":
                        pk(None)
                    else:
                        pk(ctypes_stream := _lib.stream_ptr(dev))
            with torch.cuda.stream(sa):
                vk(out, _lib.stream_ptr(dev))
                ok = torch.equal(out, ref)     # syncs stream sa only
            if not ok:
                bad += 1
                d = (out != ref)
                rows = sorted(set(d.nonzero()[:, 0].tolist()))
                print(f"   {vname} beside {pname}: launch {i}: {int(d.sum())} elements in rows {rows[:6]}", flush=True)
        torch.cuda.synchronize()
        print(f"{vname} beside {pname}: {bad} of {n} launches differ", flush=True)
