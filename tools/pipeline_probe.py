"""Debug probe: embeddings parked by the RecordStream pipeline (deferred two-lane passes + coder stream) vs a clean,
joined recomputation of the same batches; tells tower races from coder races."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf
from lossyless_amd.compressor import SyntheticImages, RecordStream

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
gen = sys.argv[2] if len(sys.argv) > 2 else "fresh"     # fresh: a new image batch per push; same: one resident batch
comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
ds = SyntheticImages(N)
saved = []
orig = RecordStream._collect


def collect(self):
    if self._pending is not None:
        payload, total_host, done, refs = self._pending
        done.synchronize()
        zb = refs[1]
        saved.append(zb[:sum(t.shape[0] for t in refs[0])].clone())
    orig(self)


RecordStream._collect = collect
stream = comp.record_stream()
x0 = ds.device_batch(0, 1024, "cuda")
for i in range(0, N, 1024):
    stream.push(ds.device_batch(i, min(i + 1024, N), "cuda") if gen == "fresh" else x0)
body = stream.finish()
z_pipe = torch.cat(saved)
print("parked rows", z_pipe.shape[0], flush=True)
bad = []
zc0 = comp.clip(x0).clone()
for i in range(0, N, 1024):
    hi = min(i + 1024, N)
    zc = comp.clip(ds.device_batch(i, hi, "cuda")) if gen == "fresh" else zc0[:hi - i]
    d = (zc != z_pipe[i:hi]).any(dim=1).nonzero().flatten().tolist()
    for r in d:
        diff = (zc[r].float() - z_pipe[i + r].float())
        bad.append((i // 1024, r, int((diff != 0).sum()), float(diff.abs().max())))
print("rows whose parked embedding differs from the clean pass:", len(bad), bad[:20], flush=True)
