"""Probe: host time to ENQUEUE one tower pass (one C call, ~90 kernel launches) vs its GPU time.
usage: python tools/enqueue_probe.py [batch=1024] [iters=20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    net = VisionTransformer(synthetic_vit_state_dict(1)).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 224, 224, 3, generator=g, device="cuda").half()
    out = torch.empty(B, 512, dtype=torch.float16, device="cuda")
    for deferred in (False, True):
        for _ in range(3):
            net(x, out=out, deferred=deferred)
        net.join()
        torch.cuda.synchronize()
        enq = []
        t0 = time.perf_counter()
        for _ in range(iters):
            t1 = time.perf_counter()
            net(x, out=out, deferred=deferred)
            enq.append(time.perf_counter() - t1)
        net.join()
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        enq.sort()
        print(f"deferred={deferred}: enqueue per pass median {1e3 * enq[len(enq) // 2]:.2f} ms (min {1e3 * enq[0]:.2f}, max "
              f"{1e3 * enq[-1]:.2f}); wall per pass {1e3 * tot / iters:.2f} ms")


if __name__ == "__main__":
    main()
