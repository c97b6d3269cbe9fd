"""Probe: is the end-to-end rate stable inside one process / across processes?  Runs R rounds of K RecordStream
steps in one process and prints each round's rate (plus the device addresses of the buffers involved).
usage: python tools/mode_probe.py [rounds=8] [steps=32] [batch=1024]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    comp, _ = hubconf.clip_compressor_b005(device="cuda:0", clip_weights="synthetic")
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, 224, 224, 3, generator=g, device="cuda").half()
    st = comp.record_stream(16)
    for _ in range(4):
        st.push(x)
    st.finish()
    rates, enq = [], []
    for r in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            st.push(x)
        t_enq = time.perf_counter() - t0
        st.finish()
        torch.cuda.synchronize()
        rates.append(B * K / (time.perf_counter() - t0))
        enq.append(1e3 * t_enq / K)
    ws = comp.clip._ws
    print("rates k img/s:", " ".join(f"{v / 1e3:.1f}" for v in rates),
          "| host enqueue ms/step:", " ".join(f"{v:.2f}" for v in enq),
          "| ws %#x x %#x" % (ws.data_ptr(), x.data_ptr()))


if __name__ == "__main__":
    main()
