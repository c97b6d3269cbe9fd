"""Probe (host only): DataLoader collation of float32 [3,224,224] items vs torch intra-op thread count."""
import sys
import time

import torch

base = torch.randn(64, 3, 224, 224)


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return 512

    def __getitem__(self, i):
        return base[i % 64], i % 10


for nt in (int(a) for a in (sys.argv[1:] or ["256", "16", "4"])):
    torch.set_num_threads(nt)
    t = time.perf_counter()
    for x, y in torch.utils.data.DataLoader(DS(), batch_size=128, num_workers=0):
        pass
    a = time.perf_counter() - t
    t = time.perf_counter()
    for x, y in torch.utils.data.DataLoader(DS(), batch_size=128, num_workers=0):
        x = x.half()
    b = time.perf_counter() - t
    print(f"threads={nt}: DataLoader {512 / a:.0f} img/s; + half {512 / b:.0f}", flush=True)
