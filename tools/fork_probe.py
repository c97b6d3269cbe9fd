"""Where do the 15 s of DataLoader worker start-up in a process that holds a GPU context go?
(bench.py reference_call_stage.)  Times os.fork() alone, then DataLoader iterator creation / first batch /
shutdown at 1, 4, 16 workers, before and after the GPU context exists."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from workloads import Stl10Shaped                      # noqa: E402
from lossyless_amd.preprocess import RawRGB, ragged_collate   # noqa: E402


def fork_once():
    t = time.perf_counter()
    pid = os.fork()
    if pid == 0:
        os._exit(0)
    t1 = time.perf_counter()
    os.waitpid(pid, 0)
    return t1 - t, time.perf_counter() - t


def loader(ds, workers):
    from torch.utils.data import DataLoader
    t0 = time.perf_counter()
    it = iter(DataLoader(ds, batch_size=128, num_workers=workers, collate_fn=ragged_collate))
    t1 = time.perf_counter()
    next(it)
    t2 = time.perf_counter()
    n = 128
    for x, y in it:
        n += len(x)
    t3 = time.perf_counter()
    del it
    t4 = time.perf_counter()
    return dict(workers=workers, create=round(t1 - t0, 3), first=round(t2 - t1, 3), rest=round(t3 - t2, 3),
                images=n, shutdown=round(t4 - t3, 3))


def main():
    ds = Stl10Shaped(8192, RawRGB())
    print("no GPU context: fork", [round(v, 4) for v in fork_once()], flush=True)
    print(loader(ds, 4), flush=True)
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    print("GPU context, nothing allocated: fork", [round(v, 4) for v in fork_once()], flush=True)
    print(loader(ds, 4), flush=True)
    import hubconf
    comp, tr = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=True)
    comp(torch.zeros(8, 3, 224, 224, device="cuda", dtype=torch.float16))
    torch.cuda.synchronize()
    print("compressor built: fork", [round(v, 4) for v in fork_once()], flush=True)
    for w in (1, 4, 16):
        print(loader(ds, w), flush=True)
    x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
    print("8 GiB more allocated: fork", [round(v, 4) for v in fork_once()], flush=True)
    print(loader(ds, 4), flush=True)


if __name__ == "__main__":
    main()
