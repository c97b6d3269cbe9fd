"""Tower throughput by slice size (one stream): the persistent GEMMs quantise to whole rounds of 320-row tiles over the
CUs -- 160 row tiles (1024 images) leave 16 of 256 CUs idle, 170 (1088 images) fill them.
    LLA_VIT_CHUNK=4352 python tools/slice_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf
from lossyless_amd.compressor import SyntheticImages
comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic",
                                       vit_chunk=int(os.environ.get("LLA_VIT_CHUNK", "4352")))
ds = SyntheticImages(20000)
for B in (int(v) for v in os.environ.get("PROBE_SIZES", "1024,1088,2176,4352").split(",")):
    x = ds.device_batch(0, B, "cuda")
    for _ in range(3):
        comp.clip(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = max(8, 16384 // B)
    for _ in range(n):
        comp.clip(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:5d}: {dt*1e3:7.3f} ms  {B/dt:9.0f} img/s", flush=True)
