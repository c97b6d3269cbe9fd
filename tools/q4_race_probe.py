"""Does a GEMM's output ever change between launches on the same operands while ANOTHER PROCESS keeps the chip busy?
(Two processes on one GPU = two hardware queues: the configuration in which test_config3's 1-rank and 2-rank files differed
by one record in 10^6 once in a while, DESIGN.md 5.3 / round 4 late.)  Every process loops lla_gemm_f16 on fixed operands and
compares a checksum of the output with the first launch's, on the device.
usage (GPU box): python tools/q4_race_probe.py [procs=2] [iters=1500] [M=217600]; env LLA_Q4_PIPE / LLA_GEMM_Q4 select the kernel"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("qkv f16", 2304, 768, 0), ("fc1 gelu", 3072, 768, 1), ("out resid", 768, 768, 2), ("fc2 resid", 768, 3072, 2)]


def child(iters, M, rank):
    import torch
    from lossyless_amd import _lib
    L = _lib.lib()
    only = os.environ.get("RACE_SHAPES")
    for name, N, K, epi in SHAPES:
        if only and name.split()[0] not in only.split(","):
            continue
        g = torch.Generator(device="cuda").manual_seed(N + K + epi)
        A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda")
        C0 = torch.randn(M, N, generator=g, device="cuda") if epi == 2 else None
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 2 else torch.float16, device="cuda")
        st = _lib.stream_ptr()
        ref, bad = None, torch.zeros((), dtype=torch.int64, device="cuda")
        for it in range(iters):
            if epi == 2:
                C.copy_(C0)
            else:
                C.zero_()
            rc = L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, st)
            assert rc == 0
            bits = C.view(torch.int16 if epi != 2 else torch.int32)
            s = bits.sum(dtype=torch.int64) + (bits[::7].to(torch.int64) * 3).sum()
            if ref is None:
                ref = s.clone()
            bad += (s != ref).to(torch.int64)
        torch.cuda.synchronize()
        print(f"rank {rank} {name}: {int(bad)} of {iters} launches differ from the first", flush=True)
        del A, W, C, C0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 217600
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(iters), str(M), str(r)]) for r in range(procs)]
    for p in ps:
        p.wait()


if __name__ == "__main__":
    main()
