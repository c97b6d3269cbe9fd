"""Probe: compress_dataset over a map-style dataset that yields what the reference's transform yields
(float32 [3,224,224] per image, hub/compressor.py:155,186), DataLoader batch 128.  Where does the time go?
usage: python tools/f32_dataset_probe.py [n=8192] [batch=128] [workers=0]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    comp, _ = hubconf.clip_compressor_b005(device="cuda:0", clip_weights="synthetic")
    g = torch.Generator().manual_seed(0)
    base = torch.randn(256, 3, 224, 224, generator=g)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            return base[i % 256], i % 10

    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "f32probe.bin")
    kw = dict(batch_size=bs, num_workers=workers)
    comp.compress_dataset(torch.utils.data.Subset(DS(), range(2 * bs)), path, kwargs_dataloader=kw, is_info=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comp.compress_dataset(DS(), path, kwargs_dataloader=kw, is_info=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the host side alone: iterate the DataLoader, nothing else
    t1 = time.perf_counter()
    for x, y in torch.utils.data.DataLoader(DS(), **kw):
        pass
    host = time.perf_counter() - t1
    t2 = time.perf_counter()
    for x, y in torch.utils.data.DataLoader(DS(), **kw):
        x = x.half()
    host_half = time.perf_counter() - t2
    print(f"n={n} batch={bs} workers={workers}: compress_dataset {n / el:.0f} img/s | DataLoader alone {n / host:.0f} img/s | "
          f"DataLoader + host .half() {n / host_half:.0f} img/s")


if __name__ == "__main__":
    main()
