"""Debug probe: records of SyntheticImages [0, N) coded in one go vs in two halves vs again (determinism)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf
from lossyless_amd import _lib
from lossyless_amd.compressor import SyntheticImages

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")


def records(lo, hi, bs=1024):
    ds = SyntheticImages(N)
    stream = comp.record_stream()
    for i in range(lo, hi, bs):
        stream.push(ds.device_batch(i, min(i + bs, hi), "cuda"))
    body = stream.finish()
    n = hi - lo
    blob = np.concatenate([np.frombuffer(int(n).to_bytes(4, "big"), np.uint8), body])
    off = np.zeros(n + 1, np.uint64)
    cnt = ctypes.c_uint32(0)
    _lib.check(_lib.lib().lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                              off.ctypes.data_as(ctypes.c_void_p), off.size, ctypes.byref(cnt)), "idx")
    return body, off


def diff(a, b, tag):
    (ba, oa), (bb, ob) = a, b
    bad = []
    for i in range(len(oa) - 1):
        ra = ba[int(oa[i]):int(oa[i + 1])]
        rb = bb[int(ob[i]):int(ob[i + 1])]
        if len(ra) != len(rb) or not np.array_equal(ra, rb):
            bad.append(i)
    print(tag, "differing images:", len(bad), bad[:20], flush=True)
    return bad


full = records(0, N)
full2 = records(0, N)
diff(full, full2, "full vs full again:")
h = N // 2
a, b = records(0, h), records(h, N)
body = np.concatenate([a[0], b[0]])
off = np.concatenate([a[1], b[1][1:] + a[1][-1]])
bad = diff(full, (body, off), "full vs halves:")
if bad:
    ds = SyntheticImages(N)
    i = bad[0]
    z1 = comp.clip(ds.device_batch(i - i % 1024, i - i % 1024 + 1024, "cuda"))[i % 1024]
    z2 = comp.clip(ds.device_batch(i, i + 1, "cuda"))[0]
    print("image", i, "embedding max abs diff batch-vs-alone", float((z1.float() - z2.float()).abs().max()))
