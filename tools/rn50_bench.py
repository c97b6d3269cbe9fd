"""RN50-CLIP tower micro-benchmark (run on the GPU box; wrap in rocprofv3 --kernel-trace --stats for a kernel table).
usage: python tools/rn50_bench.py [batch=256] [chunk=64] [iters=5]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    net = ModifiedResNet(synthetic_rn50_state_dict(1), chunk=chunk).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 224, 224, 3, generator=g, device="cuda").half()
    net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        net(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"B={B} chunk={chunk}: {dt * 1e3:.2f} ms  {B / dt:.0f} img/s")


if __name__ == "__main__":
    main()
