"""Entropy-stage micro-benchmark (bench.py's entropy_stage / hyperprior legs alone).
usage (GPU box): python tools/entropy_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import hubconf  # noqa: E402

comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
if not (len(sys.argv) > 1 and sys.argv[1] == "--raw-decode"):
    e = bench.entropy_stage_leg(comp, "cuda")
    print({k: e[k] for k in ("img_per_sec", "decode_img_per_sec", "bits_per_img")})
    h = bench.hyperprior_leg("cuda")
    print({k: h[k] for k in ("encode_rows_per_sec", "decode_rows_per_sec", "encode_ms", "decode_ms")})


def raw_decode_timing(B=1024, iters=20):
    """Decode timing without the equality check (timing only)."""
    import numpy as np
    import torch
    from lossyless_amd import _lib
    t = comp._tables()
    rng = np.random.default_rng(2)
    z = torch.from_numpy(rng.normal(size=(B, 512)).astype(np.float32) * 0.3).cuda()
    eb = comp.entropy_bottleneck
    payload, offsets, _ = eb.encode_device(z, t, record_prefix=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for B_ in (B,):
        eb.decode_device(payload, offsets, B_, t, record_prefix=True)
        e0.record()
        for _ in range(iters):
            eb.decode_device(payload, offsets, B_, t, record_prefix=True)
        e1.record()
        torch.cuda.synchronize()
        print(f"raw decode B={B_}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per batch")


if len(sys.argv) > 1 and sys.argv[1] == "--raw-decode":
    raw_decode_timing()
    raw_decode_timing(B=16384)
