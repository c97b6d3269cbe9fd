"""Generate the committed golden vectors under tests/golden/ (run in the build container).

They are produced by the CPU oracle (oracle/), because the reference's own implementation
of this path (compressai / clip) is not importable here: they pin the
documented algorithm and self-consistency, and are labelled as such.

  symbols_<beta>.npy   int32 [64,512]  symbols sampled from the model pmf, escapes forced
  golden_<beta>.bin    the reference container (hub/compressor.py:192-196) of their streams
  vit_synth_z.npy      fp32 [4,512]    fp32-CPU tower output, synthetic seed-1 weights,
                                       images = seeded uint8 (see tests/test_gpu_vit.py)
  gaussian_golden.npz  GaussianConditional (64-level scale table): symbols / table rows int32 [16,96],
                       the scale-table CDF digest and the container of their strings
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BETAS, GOLDEN, load_tables, sample_symbols  # noqa: E402
from oracle import cbind, container, vit  # noqa: E402
from lossyless_amd.clip_vit import synthetic_vit_state_dict  # noqa: E402


def synth_images(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    return ((u8.float() / 255 - mean) / std).half()  # NHWC fp16


def main():
    for i, tag in enumerate(BETAS):
        tab = load_tables(tag)
        sym = sample_symbols(tab, 64, seed=100 + i, escape_boost=0.01)
        sym[0, :4] = [2 ** 20, -2 ** 20, 2 ** 29, -2 ** 29]  # 6-8 digit payloads
        np.save(os.path.join(GOLDEN, f"symbols_{tag}.npy"), sym)
        strings = [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"]) for s in sym]
        container.write_container(os.path.join(GOLDEN, f"golden_{tag}.bin"), strings)
        print(tag, "mean bytes", np.mean([len(s) for s in strings]))
    from oracle import gc
    import hashlib
    tab = gc.derive_tables(gc.get_scale_table())
    rng = np.random.default_rng(77)
    idx = rng.integers(0, 64, size=(16, 96)).astype(np.int32)
    sym = np.rint(rng.normal(size=idx.shape) * tab["scale_table"][idx] * 1.2).astype(np.int32)
    sym[0, :6] = [2 ** 30, -(2 ** 30), 10 ** 5, -4000, 17, -17]
    idx[0, :6] = [0, 63, 0, 5, 0, 1]
    strings = gc.compress(sym, idx, tab)
    blob = np.frombuffer(container.container_bytes(strings), dtype=np.uint8)
    digest = hashlib.sha256(tab["cdf"].tobytes() + tab["cdf_len"].tobytes() + tab["offset"].tobytes()).hexdigest()
    np.savez(os.path.join(GOLDEN, "gaussian_golden.npz"), symbols=sym, indexes=idx, container=blob,
             table_sha256=np.array(digest))
    print("gaussian", len(blob), digest[:16])
    x = synth_images(4).permute(0, 3, 1, 2).float()
    z = vit.vit_b32_forward(synthetic_vit_state_dict(1), x).numpy()
    np.save(os.path.join(GOLDEN, "vit_synth_z.npy"), z.astype(np.float32))
    print("vit z", z.shape, float(np.abs(z).mean()), float(z.std()))


if __name__ == "__main__":
    main()
