"""Board power and shader clock, read from the SMI while one kernel runs back to back (GPU box).
docs/history/DESIGN_rounds_1-5.md 5.6 infers from rocprofv3 counters (GRBM_GUI_ACTIVE / duration) that the tower's GEMMs run at 1.3-1.5 GHz
"under the power limit"; this probe reads the limit, the average socket power and the clocks directly while
  * the four-wave GEMM (QKV shape, M = 217 600, pipelined and serial epilogue),
  * hipBLASLt's GEMM of the same shape (torch.matmul),
  * a memory-bound kernel (fp32 copy of 1.3 GB)
each run for a few seconds.  usage: python tools/power_probe.py [seconds=4]"""
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi_once():
    try:
        p = subprocess.run(["rocm-smi", "-P", "-c", "-M", "--json"], capture_output=True, text=True, timeout=20)
        return p.stdout.strip()
    except Exception as e:  # noqa: BLE001
        return "ERR " + repr(e)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            self.samples.append((time.time(), smi_once()))


def summarise(tag, samples, t0, t1, work_note):
    pw, sclk, cap = [], [], None
    for t, s in samples:
        if t < t0 + 0.7 or t > t1:      # (skip the ramp)
            continue
        try:
            d = json.loads(s)
        except Exception:  # noqa: BLE001
            continue
        card = d.get("card0", d[next(iter(d))])
        for k, v in card.items():
            kl = k.lower()
            m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
            if not m:
                continue
            x = float(m.group(0))
            if "power" in kl and "max" in kl:
                cap = x
            elif "power" in kl and ("average" in kl or "current" in kl or "socket" in kl):
                pw.append(x)
            elif "sclk" in kl and "level" in kl:
                sclk.append(x)
    mean = lambda a: sum(a) / len(a) if a else float("nan")  # noqa: E731
    print(f"{tag:>34}: power {mean(pw):7.1f} W (max {max(pw) if pw else float('nan'):7.1f}, cap {cap}), "
          f"sclk {mean(sclk):6.0f} MHz, {len(pw)} samples | {work_note}")


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    import torch
    from lossyless_amd import _lib
    L = _lib.lib()
    print("raw sample:", smi_once()[:1500])
    M, N, K = 217600, 2304, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda")
    C = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    src = torch.randn(M * 768 * 2, device="cuda")
    dst = torch.empty_like(src)
    st = _lib.stream_ptr()

    def ours():
        L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, 0, st)

    def blas():
        torch.matmul(A, W.t(), out=C)

    def copy():
        dst.copy_(src)

    which = os.environ.get("POWER_PROBE", "ours,blas,copy,idle").split(",")
    for tag, fn, flop, nbytes in [("ours", ours, 2.0 * M * N * K, 0), ("blas", blas, 2.0 * M * N * K, 0),
                                  ("copy", copy, 0, 2.0 * src.numel() * 4), ("idle", None, 0, 0)]:
        if tag not in which:
            continue
        sm = Sampler()
        if fn:
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        sm.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < secs:
            if fn:
                for _ in range(50):
                    fn()
                n += 50
                torch.cuda.synchronize()
            else:
                time.sleep(0.2)
        e1.record()
        torch.cuda.synchronize()
        t1 = time.time()
        sm.stop = True
        sm.join()
        ms = e0.elapsed_time(e1) / max(n, 1)
        note = (f"{ms * 1e3:7.1f} us per launch" + (f", {flop / ms / 1e9:7.1f} TFLOP/s" if flop else "") +
                (f", {nbytes / ms / 1e6:7.1f} GB/s" if nbytes else "")) if fn else ""
        summarise(f"{tag} (LLA_GEMM_W8={os.environ.get('LLA_GEMM_W8', '1')})" if tag == "ours" else tag,
                  sm.samples, t0, t1, note)


if __name__ == "__main__":
    main()
