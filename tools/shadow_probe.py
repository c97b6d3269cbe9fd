"""Debug probe (ablation build): shadow execution of every tower kernel inside the two-lane deferred pipeline; prints
which kernel's two executions disagreed.
    LLA_LIB=lossyless_amd/liblossyless_amd_ablation.so python tools/shadow_probe.py [batches]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf
from lossyless_amd.compressor import SyntheticImages

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = int(os.environ.get("PROBE_BATCH", "1024"))
comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
x = SyntheticImages(4096).device_batch(0, B, "cuda")
ROWS = B * 50
per_lane = ROWS * (768 * 4 + 768 * 2 + 3072 * 2)
shadow = torch.zeros(2 * per_lane, dtype=torch.uint8, device="cuda")
log = torch.zeros(1 + 4 * 255, dtype=torch.int64, device="cuda")
os.environ["LLA_VIT_SNAPSHOT_ROWS"] = str(ROWS)
os.environ["LLA_VIT_SHADOW"] = str(shadow.data_ptr())
os.environ["LLA_VIT_SHADOW_LOG"] = str(log.data_ptr())
from lossyless_amd.compressor import RecordStream
saved = []
orig = RecordStream._collect


def collect(self):
    if self._pending is not None:
        payload, total_host, done, refs = self._pending
        done.synchronize()
        saved.append(refs[1][:sum(t.shape[0] for t in refs[0])].clone())
    orig(self)


RecordStream._collect = collect
stream = comp.record_stream()
for i in range(n_batches):
    stream.push(x)
stream.finish()
torch.cuda.synchronize()
z_pipe = torch.cat(saved).view(n_batches, B, 512)
os.environ.pop("LLA_VIT_SHADOW", None)
zc = comp.clip(x)
bad = (z_pipe != zc[None]).any(dim=2).nonzero().tolist()
print(f"embeddings differing from the clean pass: {len(bad)} {bad[:12]}", flush=True)
lg = log.cpu().numpy()
kinds = ["ln_1", "qkv gemm", "attention", "out_proj gemm (+=)", "ln_2", "c_fc gemm", "c_proj gemm (+=)"]
widths = [768 * 2, 2304 * 2, 768 * 2, 768 * 4, 768 * 2, 3072 * 2, 768 * 4]     # bytes per row
print(f"{int(lg[0])} mismatching 16-byte pieces over {n_batches} batches of {B}", flush=True)
from collections import Counter
c = Counter()
for k in range(min(int(lg[0]), 255)):
    tag, idx, a, b = (int(v) for v in lg[1 + 4 * k:5 + 4 * k])
    layer, kind = tag // 8, tag % 8
    row, col = (idx * 16) // widths[kind], ((idx * 16) % widths[kind])
    c[(layer, kinds[kind], row)] += 1
for (layer, kind, row), n in sorted(c.items()):
    print(f"  layer {layer} {kind}: row {row} (image {row // 50}, token {row % 50}): {n} pieces", flush=True)
