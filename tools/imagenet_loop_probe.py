"""Where does the main process spend its time in the ImageNet-shaped reference call (gpu_preprocess=True, workers)?"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from workloads import MixedSizeImages                  # noqa: E402
import hubconf                                          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
comp, tr = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=True)
ds = MixedSizeImages(n, tr)
kw = dict(batch_size=512, num_workers=workers)
for rep in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    comp.compress_dataset(ds, "/tmp/p.bin", label_file="/tmp/p.npy", kwargs_dataloader=kw, is_info=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print(f"{workers} workers: {n} images in {el:.2f} s = {n / el:.0f} img/s", flush=True)
cProfile.run('comp.compress_dataset(ds, "/tmp/p.bin", label_file="/tmp/p.npy", kwargs_dataloader=kw, is_info=False)', "/tmp/prof")
pstats.Stats("/tmp/prof").sort_stats("tottime").print_stats(14)
# the dataset's own cost per image, one process
t = time.perf_counter()
for i in range(512):
    ds[i]
print(f"dataset[i] (generate + Image.fromarray + RawRGB): {(time.perf_counter() - t) / 512 * 1e3:.3f} ms per image")
