#!/usr/bin/env python3
"""How long should a column tile of an EPI_RESID_LNX GEMM wait for its two siblings (gemm_q4.hip, `lnx_wait` shader
cycles)?  One 8704-image tower pass per call, the wait budget switched through `lla_tower_set_option`
between calls, the settings interleaved over several rounds on one box; prints ms per pass and the embeddings' sha per
setting (every setting gives the same bits).

  python tools/lnx_wait_sweep.py [--rounds 4] [--passes 6]
"""
import argparse
import hashlib
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--images", type=int, default=8704)
    ap.add_argument("--waits", default="", help="comma list: time only these wait budgets (for A/B runs of library builds: LLA_LIB=...)")
    args = ap.parse_args()
    from lossyless_amd import _lib
    from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict
    tower = VisionTransformer(synthetic_vit_state_dict(1)).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(args.images, 224, 224, 3, generator=g, device="cuda").half()
    T = _lib.Tower
    tower(x)
    h = tower.tower(x.device)
    settings = [("LayerNorm kernels (LNX off)", {T.OPT_LNX: 0}), ("every row tile by the clean-up kernel", {T.OPT_LNX: 1, T.OPT_LNX_WAIT: -1}),
                ("poll once", {T.OPT_LNX: 1, T.OPT_LNX_WAIT: 0})] + \
               [(f"wait {w}", {T.OPT_LNX: 1, T.OPT_LNX_WAIT: w}) for w in (1500, 3000, 6000, 12000, 24000, 100000, 1 << 24)]
    if args.waits:
        settings = [(f"wait {int(w)}", {T.OPT_LNX: 1, T.OPT_LNX_WAIT: int(w)}) for w in args.waits.split(",")]
    ms = {name: [] for name, _ in settings}
    sha = {}
    for r in range(args.rounds):
        for name, opts in settings:
            for k, v in opts.items():
                h.set_option(k, v)
            z = tower(x)                                   # (first pass of a setting untimed)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.passes):
                z = tower(x)
            b.record()
            torch.cuda.synchronize()
            ms[name].append(a.elapsed_time(b) / args.passes)
            sha.setdefault(name, set()).add(hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:12])
    h.set_option(T.OPT_LNX, 1)
    h.set_option(T.OPT_LNX_WAIT, 24000)
    print(f"library {os.environ.get('LLA_LIB', 'product')}: {args.images} images per pass, {args.passes} timed passes per cell, {args.rounds} interleaved rounds; ms per pass")
    for name, _ in settings:
        v = ms[name]
        print(f"{name:40s} " + " ".join(f"{t:7.2f}" for t in v) + f"   median {sorted(v)[len(v) // 2]:7.2f}   {args.images / sorted(v)[len(v) // 2] * 1e3:9.0f} img/s   sha {','.join(sorted(sha[name]))}")
    assert len(set.union(*sha.values())) == 1, "settings differ in bits"


if __name__ == "__main__":
    main()
