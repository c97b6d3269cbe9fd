"""cProfile of the main process during the reference call with 16 workers, in a process shaped like bench.py's
(cpu_baseline has run, a 32768-image dataset is resident)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from workloads import Stl10Shaped                      # noqa: E402
import hubconf                                          # noqa: E402
import bench                                            # noqa: E402

if "--baseline" in sys.argv:
    bench.cpu_baseline(min_seconds=2.0, max_seconds=4.0)
full = Stl10Shaped(32768, None)
kw = dict(batch_size=128, num_workers=16)
for gpu_pre in (False, True):
    comp, tr = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=gpu_pre)
    full.transform = tr
    sub = torch.utils.data.Subset(full, range(512 if not gpu_pre else 4096))
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        comp.compress_dataset(sub, "/tmp/p.bin", label_file="/tmp/p.npy", kwargs_dataloader=kw, is_info=False)
        torch.cuda.synchronize()
        print("gpu_preprocess", gpu_pre, "rep", rep, round(time.perf_counter() - t, 3), flush=True)
    cProfile.run('comp.compress_dataset(sub, "/tmp/p.bin", label_file="/tmp/p.npy", kwargs_dataloader=kw, is_info=False)',
                 "/tmp/prof")
    pstats.Stats("/tmp/prof").sort_stats("tottime").print_stats(10)
