"""Which stage of bench.py makes fork() (hence DataLoader worker start-up) slow in the same process?
Prints the time of one bare fork() after each stage; pinned host memory is the known cause (copied eagerly)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import hubconf                                          # noqa: E402
import bench                                            # noqa: E402


_buf = None


def fork_ms(tag, forks=4):
    """fork `forks` children (kept alive), then time the first H2D copies: the GPU stall the forks leave behind."""
    global _buf
    import signal
    torch.cuda.synchronize()
    if _buf is None:
        _buf = torch.empty(4 << 20, dtype=torch.uint8).pin_memory()
    s = torch.cuda.Stream()
    t0 = time.perf_counter()
    pids = []
    for _ in range(forks):
        pid = os.fork()
        if pid == 0:
            time.sleep(30)
            os._exit(0)
        pids.append(pid)
    t1 = time.perf_counter()
    worst = 0.0
    for _ in range(5):
        t = time.perf_counter()
        with torch.cuda.stream(s):
            _buf.to("cuda", non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s)
        ev.synchronize()
        worst = max(worst, time.perf_counter() - t)
    for p in pids:
        os.kill(p, signal.SIGKILL)
        os.waitpid(p, 0)
    st = torch.cuda.host_memory_stats() if hasattr(torch.cuda, "host_memory_stats") else {}
    print(f"{tag}: {forks} forks {1e3 * (t1 - t0):.0f} ms, worst H2D copy after them {1e3 * worst:.1f} ms; torch pinned "
          f"{st.get('allocated_bytes.current', 0) >> 20} MiB; device reserved {torch.cuda.memory_reserved() >> 20} MiB",
          flush=True)


torch.zeros(1, device="cuda")
print("HSA_USERPTR_FOR_PAGED_MEM =", os.environ.get("HSA_USERPTR_FOR_PAGED_MEM"))
fork_ms("start")
base = bench.cpu_baseline(min_seconds=2.0, max_seconds=4.0)
fork_ms("cpu_baseline")
comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
fork_ms("compressor built")
if "--short" in sys.argv:
    sys.exit(0)
from lossyless_amd.compressor import SyntheticImages
comp.compress_dataset(SyntheticImages(70000), "/tmp/q.bin", is_info=False)
fork_ms("70k synthetic images")
x = bench.synth_batch(1024, 0, "cuda")
bench.verify_first_batch(comp, x)
fork_ms("verify_first_batch")
bench.entropy_stage_leg(comp, "cuda")
fork_ms("entropy leg")
bench.preprocess_leg(comp, "cuda")
fork_ms("preprocess leg")
bench.hyperprior_leg("cuda")
fork_ms("hyperprior leg")
bench.stl10_shaped_leg(comp, "cuda")
fork_ms("stl10 leg")
c2, tr = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=False)
fork_ms("second compressor")
