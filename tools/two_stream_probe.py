"""Probe: does running two half-batches of the tower on two HIP streams beat one full batch on one stream?
(tails of the persistent GEMMs and the HBM-bound LayerNorm / attention kernels of one half can run beside the
other half's GEMMs).  usage: python tools/two_stream_probe.py [batch=1024] [steps=20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sd = synthetic_vit_state_dict(1)
    nets = [VisionTransformer(sd).cuda() for _ in range(4)]
    g = torch.Generator(device="cuda").manual_seed(0)
    X = torch.randn(B, 224, 224, 3, generator=g, device="cuda").half()

    def run(parts):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        xs = X.chunk(parts)
        outs = [torch.empty(x.shape[0], 512, dtype=torch.float16, device="cuda") for x in xs]

        def step():
            for s, net, x, o in zip(streams, nets, xs, outs):
                with torch.cuda.stream(s):
                    net(x, out=o)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return dt, torch.cat(outs)

    def run_full(n):   # n streams, each with its own full batch
        streams = [torch.cuda.Stream() for _ in range(n)]
        outs = [torch.empty(B, 512, dtype=torch.float16, device="cuda") for _ in range(n)]

        def step():
            for s, net, o in zip(streams, nets, outs):
                with torch.cuda.stream(s):
                    net(X, out=o)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return dt, outs

    for n in (1, 2, 3, 2):
        dt, outs = run_full(n)
        print(f"{n} stream(s) x {B} images each: {dt * 1e3:7.3f} ms per round = {n * B / dt:9.0f} img/s  "
              f"all equal={all(bool((o == outs[0]).all()) for o in outs)}")

    ref = None
    for parts in (1, 2, 1, 2):
        dt, z = run(parts)
        if ref is None:
            ref = z
        same = bool((z == ref).all())
        print(f"{parts} stream(s) x {B // parts} images: {dt * 1e3:7.3f} ms per {B} images = {B / dt:9.0f} img/s  bit-identical={same}")


if __name__ == "__main__":
    main()
