"""How long does one H2D copy out of a pinned buffer take (side stream + event.synchronize), plain / after
MADV_DONTFORK / with forked children alive / with busy children?"""
import ctypes
import os
import signal
import sys
import time

import torch

libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]


def h2d(buf, n=20):
    s = torch.cuda.Stream()
    out = []
    for _ in range(n):
        t = time.perf_counter()
        with torch.cuda.stream(s):
            d = buf.to("cuda", non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s)
        ev.synchronize()
        out.append(time.perf_counter() - t)
    out.sort()
    return f"median {1e3 * out[len(out) // 2]:.2f} ms, max {1e3 * out[-1]:.2f} ms"


def children(k, busy):
    pids = []
    for _ in range(k):
        pid = os.fork()
        if pid == 0:
            if busy:
                while True:
                    sum(range(100000))
            else:
                time.sleep(60)
            os._exit(0)
        pids.append(pid)
    return pids


def reap(pids):
    for p in pids:
        os.kill(p, signal.SIGKILL)
        os.waitpid(p, 0)


torch.zeros(1, device="cuda")
for mb in (4, 38):
    buf = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    print(mb, "MB plain:", h2d(buf), flush=True)
    pids = children(16, False)
    print(mb, "MB, 16 idle children (buffer copied into them):", h2d(buf), flush=True)
    reap(pids)
    rc = libc.madvise(ctypes.c_void_p(buf.data_ptr()), mb << 20, 10)
    print(mb, "MB after MADV_DONTFORK rc", rc, ":", h2d(buf), flush=True)
    pids = children(16, False)
    print(mb, "MB DONTFORK, 16 idle children:", h2d(buf), flush=True)
    reap(pids)
    pids = children(16, True)
    print(mb, "MB DONTFORK, 16 busy children:", h2d(buf), flush=True)
    reap(pids)
    print(mb, "MB DONTFORK, children gone:", h2d(buf), flush=True)
    libc.madvise(ctypes.c_void_p(buf.data_ptr()), mb << 20, 11)

big = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
t = time.perf_counter()
pids = children(4, False)
print(f"4 forks with 1 GiB pinned (HSA_USERPTR_FOR_PAGED_MEM={os.environ.get('HSA_USERPTR_FOR_PAGED_MEM')}): "
      f"{time.perf_counter() - t:.3f} s;", "first copies after:", h2d(buf, 5), flush=True)
reap(pids)
t = time.perf_counter()
d = big.to("cuda", non_blocking=True)
torch.cuda.synchronize()
print(f"1 GiB H2D: {1.0737 / (time.perf_counter() - t):.1f} GB/s")
