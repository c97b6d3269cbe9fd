#!/usr/bin/env python
"""Pin the oracle (and through it the HIP path) against the REAL third-party code, wherever that
code is installable.

The arithmetic of the compress_dataset path lives in compressai==1.1.5 and clip==1.0
(/root/reference/requirements/environment.yaml:98,105), neither of which is vendored in the
reference nor importable in the build container -- so every committed fixture is
"oracle-generated" and parity is UNPINNED against live third-party code.  This script closes
that gap on any machine where `pip install compressai==1.1.5` (and optionally
`git+https://github.com/openai/CLIP.git`) works.  It needs no GPU.  It

  1. builds compressai's EntropyBottleneck(512, init_scale=10, filters=[3,3,3,3]) exactly as
     hub/compressor.py:49-63 does, loads each shipped checkpoint, runs update(force=True) and
     diffs `_quantized_cdf / _cdf_length / _offset` with tests/golden/tables_<beta>.npz;
  2. runs compressai._CXX.pmf_to_quantized_cdf on the KAT pmfs of tests/test_oracle.py and on
     random pmfs and diffs with oracle.cbind.pmf_to_quantized_cdf and lla_pmf_to_quantized_cdf;
  3. codes tests/golden/symbols_<beta>.npy with ans.RansEncoder().encode_with_indexes (one call
     per image, like EntropyModel.compress) and diffs the container with
     tests/golden/golden_<beta>.bin; decodes it back with ans.RansDecoder().decode_with_indexes;
  4. (with --write) REGENERATES the fixtures from compressai so that they become reference-made;
  5. (if `clip` and $LOSSYLESS_CLIP_WEIGHTS are available) compares oracle/vit.py with
     clip's VisionTransformer on 8 seeded images (fp32, CPU).

Exit status: 0 all equal, 1 a difference was found, 77 compressai not importable (skipped).
"""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BETAS = ("1e-01", "5e-02", "1e-02")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="overwrite tests/golden/* with compressai's outputs")
    args = ap.parse_args()
    try:
        import compressai
        from compressai.entropy_models import EntropyBottleneck
        from compressai.models.utils import update_registered_buffers
        from compressai import ans
        from compressai._CXX import pmf_to_quantized_cdf as cxx_pmf_to_cdf
    except Exception as e:  # noqa: BLE001
        print(f"SKIP: compressai is not importable here ({e!r}); parity stays unpinned")
        return 77
    import torch
    from oracle import cbind
    print("compressai", getattr(compressai, "__version__", "?"))
    bad = 0

    # -- 2. pmf_to_quantized_cdf
    rng = np.random.default_rng(0)
    pmfs = [np.array([0.5, 0.25, 0.25], np.float32), np.array([1e-9, 0.5, 0.5 - 1e-9], np.float32),
            np.array([0.3, 1e-12, 1e-12, 0.7], np.float32)]
    for _ in range(500):
        n = int(rng.integers(2, 33))
        p = rng.dirichlet(np.full(n, 0.3)).astype(np.float32)
        p[rng.random(n) < 0.2] *= 1e-7
        pmfs.append(p)
    for p in pmfs:
        want = list(cxx_pmf_to_cdf(p.tolist(), 16))
        try:
            got = cbind.pmf_to_quantized_cdf(p).tolist()
        except ValueError:
            got = None
        if got != want:
            bad += 1
            print("pmf_to_quantized_cdf differs for", p.tolist()[:6], "...")
    print(f"pmf_to_quantized_cdf: {len(pmfs)} pmfs checked")

    enc, dec = ans.RansEncoder(), ans.RansDecoder()
    for tag in BETAS:
        sd = torch.load(os.path.join(ROOT, "lossyless_amd", "assets", f"beta{tag}_factorized_rate.pt"),
                        map_location="cpu", weights_only=True)
        sd = dict(sd)
        for k in ("_quantized_cdf", "_offset", "_cdf_length"):      # as the reference ships them
            sd["entropy_bottleneck." + k] = torch.IntTensor()
        # -- 1. tables, the way hub/compressor.py:49-63 builds them
        eb = EntropyBottleneck(512, init_scale=10, filters=[3, 3, 3, 3])
        update_registered_buffers(eb, "entropy_bottleneck", ["_quantized_cdf", "_offset", "_cdf_length"], sd)
        eb.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("entropy_bottleneck.")})
        eb.update(force=True)
        gold = np.load(os.path.join(GOLDEN, f"tables_{tag}.npz"))
        ref = dict(cdf=eb._quantized_cdf.numpy().astype(np.int32), cdf_len=eb._cdf_length.numpy().astype(np.int32),
                   offset=eb._offset.numpy().astype(np.int32))
        for k, v in ref.items():
            same = v.shape == gold[k].shape and np.array_equal(v, gold[k])
            if not same:
                bad += 1
                n = int((v != gold[k]).sum()) if v.shape == gold[k].shape else -1
                print(f"[{tag}] table {k} differs from the golden fixture ({n} entries)")
        # -- 3. streams
        sym = np.load(os.path.join(GOLDEN, f"symbols_{tag}.npy"))
        cdfs, lens, offs = ref["cdf"].tolist(), ref["cdf_len"].tolist(), ref["offset"].tolist()
        idx = list(range(sym.shape[1]))
        strings = [enc.encode_with_indexes(s.tolist(), idx, cdfs, lens, offs) for s in sym]
        blob = struct.pack(">I", len(strings)) + b"".join(struct.pack(">I", len(s)) + s for s in strings)
        want = open(os.path.join(GOLDEN, f"golden_{tag}.bin"), "rb").read()
        if blob != want:
            bad += 1
            print(f"[{tag}] container differs from golden_{tag}.bin ({len(blob)} vs {len(want)} bytes)")
        for s, row in zip(strings, sym):
            if dec.decode_with_indexes(s, idx, cdfs, lens, offs) != row.tolist():
                bad += 1
                print(f"[{tag}] compressai does not decode its own stream back?!")
                break
        if args.write:
            np.savez_compressed(os.path.join(GOLDEN, f"tables_{tag}.npz"), **ref,
                                median=gold["median"], exp_scale=gold["exp_scale"], bias=gold["bias"])
            open(os.path.join(GOLDEN, f"golden_{tag}.bin"), "wb").write(blob)
            print(f"[{tag}] fixtures rewritten from compressai")
        print(f"[{tag}] tables + {len(strings)} streams checked")

    # -- 5. tower
    w = os.environ.get("LOSSYLESS_CLIP_WEIGHTS")
    try:
        import clip  # noqa: F401
        have_clip = True
    except Exception:  # noqa: BLE001
        have_clip = False
    if have_clip and w:
        from oracle import vit
        from lossyless_amd.clip_vit import load_clip_visual_state_dict
        model, _ = clip.load(w, device="cpu", jit=False)
        g = torch.Generator().manual_seed(0)
        x = torch.randn(8, 3, 224, 224, generator=g)
        with torch.no_grad():
            want = model.visual(x.float()).float()
            got = vit.vit_b32_forward({k: v.float() for k, v in load_clip_visual_state_dict(w).items()}, x)
        rel = ((got - want).norm(dim=1) / want.norm(dim=1)).max().item()
        print(f"oracle tower vs clip.VisionTransformer: max rel L2 {rel:.2e}")
        bad += rel > 1e-4
    else:
        print("tower: clip / $LOSSYLESS_CLIP_WEIGHTS not available, skipped")
    print("RESULT:", "PINNED (all equal)" if not bad else f"{bad} DIFFERENCES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
