"""BASELINE.json configs[2] / configs[4] harness: rate sweep over the three shipped rate points and
round trip + downstream accuracy, on real data when it is available.

    python tools/rate_sweep.py --images X.npy [--labels Y.npy] [--test-images Xt.npy --test-labels Yt.npy]

`X.npy`: uint8 images [N, H, W, 3] (e.g. STL10 96x96 or ImageNet-val resized); they go through the GPU
preprocessing (Pillow-exact resize / centre crop / CLIP normalisation) and `compress_dataset`.  With real CLIP
weights (`$LOSSYLESS_CLIP_WEIGHTS`) the numbers are comparable with the reference's (README.md:75,
notebooks/Hub.ipynb: 1506.6 bits/img and 98.64 % LinearSVC(C=7e-3) accuracy on STL10 at beta = 5e-2).
Without `--images` the sweep runs on synthetic images and says so; without real weights it says so too:
both cases exercise the harness, not the reference's numbers (the assets are not available offline).
Prints one JSON line per rate point.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images")
    ap.add_argument("--labels")
    ap.add_argument("--test-images")
    ap.add_argument("--test-labels")
    ap.add_argument("--n-synthetic", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()

    weights = os.environ.get("LOSSYLESS_CLIP_WEIGHTS", "synthetic")
    if args.images:
        X = torch.from_numpy(np.load(args.images))
        data = os.path.basename(args.images)
    else:
        g = torch.Generator().manual_seed(0)
        X = torch.randint(0, 256, (args.n_synthetic, 96, 96, 3), generator=g, dtype=torch.uint8)
        data = f"synthetic uint8 96x96 x{args.n_synthetic} (no --images: assets absent)"
    Y = np.load(args.labels) if args.labels else None
    Xt = torch.from_numpy(np.load(args.test_images)) if args.test_images else None
    Yt = np.load(args.test_labels) if args.test_labels else None

    for name in ("clip_compressor_b01", "clip_compressor_b005", "clip_compressor_b001"):
        comp, _ = getattr(hubconf, name)(device="cuda", clip_weights=weights)
        with tempfile.TemporaryDirectory() as d:
            f = os.path.join(d, "Z.bin")
            t0 = time.perf_counter()
            comp.compress_dataset(X, f, kwargs_dataloader=dict(batch_size=args.batch), is_info=False)
            torch.cuda.synchronize()
            enc = time.perf_counter() - t0
            bits = 8 * os.path.getsize(f) / len(X)
            t0 = time.perf_counter()
            Z = comp.decompress_dataset(f, is_info=False)
            dec = time.perf_counter() - t0
            # the file decodes to exactly what compressor(X) returns
            ref = torch.cat([comp(X[i:i + args.batch].cuda()) for i in range(0, len(X), args.batch)])
            assert np.array_equal(Z, ref.cpu().numpy())
            acc = "skipped: labels / test split not given"
            if Y is not None and Xt is not None and Yt is not None:
                from sklearn.svm import LinearSVC
                ft = os.path.join(d, "Zt.bin")
                comp.compress_dataset(Xt, ft, kwargs_dataloader=dict(batch_size=args.batch), is_info=False)
                Zt = comp.decompress_dataset(ft, is_info=False)
                clf = LinearSVC(C=7e-3).fit(Z, Y)          # README.md:75
                acc = float(clf.score(Zt, Yt))
        print(json.dumps(dict(rate_point=name, data=data, clip_weights=weights, images=len(X),
                              bits_per_img=round(bits, 2), encode_img_per_sec=round(len(X) / enc, 1),
                              decode_img_per_sec=round(len(X) / dec, 1), round_trip="exact",
                              linear_svc_accuracy=acc)))


if __name__ == "__main__":
    main()
