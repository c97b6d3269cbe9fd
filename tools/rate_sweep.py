"""BASELINE.json configs[2] / configs[4] harness: rate sweep over the three shipped rate points and
round trip + downstream accuracy, on real data when it is available.

    python tools/rate_sweep.py --images X.npy [--labels Y.npy] [--test-images Xt.npy --test-labels Yt.npy]
    python tools/rate_sweep.py --stl10-shaped          # configs[4] shape: 5 000 train / 8 000 test, 96x96, 10 classes
    python tools/rate_sweep.py --imagenet-shaped 50000 # configs[2] shape: 50 000 photos of mixed sizes

`X.npy`: uint8 images [N, H, W, 3] (e.g. STL10 96x96 or ImageNet-val resized); they go through the GPU
preprocessing (Pillow-exact resize / centre crop / CLIP normalisation) and `compress_dataset`.  With real CLIP
weights (`$LOSSYLESS_CLIP_WEIGHTS`) the numbers are comparable with the reference's (README.md:75,
notebooks/Hub.ipynb: 1506.6 bits/img and 98.64 % LinearSVC(C=7e-3) accuracy on STL10 at beta = 5e-2).
The two `-shaped` modes run the REFERENCE CALL -- a torchvision-style `Dataset(transform=transform)` handed to
`compress_dataset(dataset, file, label_file, kwargs_dataloader)` (hub/compressor.py:150-207) with the
compressor built with `gpu_preprocess=True` -- on generated stand-ins of the real datasets' shapes
(tools/workloads.py: the real files cannot be fetched offline), and say so in `data`.
Without real weights the output says so too: these runs exercise the path at the datasets' scale and shapes,
not the reference's numbers.  Prints one JSON line per rate point.
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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hubconf  # noqa: E402


def _compress(comp, data, f, label_file, loader):
    t0 = time.perf_counter()
    comp.compress_dataset(data, f, label_file=label_file, kwargs_dataloader=loader, is_info=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def _represent(comp, data, idx, batch):
    """compressor(X) for the samples `idx` of a tensor or a dataset of raw images."""
    out = []
    for i in range(0, len(idx), batch):
        j = idx[i:i + batch]
        if isinstance(data, torch.Tensor):
            x = data[j].cuda()
        else:
            x = [data[int(k)][0] for k in j]          # raw uint8 [H,W,3] tensors of any size
        out.append(comp(x).cpu())
    return torch.cat(out).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images")
    ap.add_argument("--labels")
    ap.add_argument("--test-images")
    ap.add_argument("--test-labels")
    ap.add_argument("--stl10-shaped", action="store_true")
    ap.add_argument("--imagenet-shaped", type=int, default=0, metavar="N")
    ap.add_argument("--n-synthetic", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--workers", type=int, default=16, help="DataLoader workers of the -shaped modes")
    ap.add_argument("--check", type=int, default=4096,
                    help="-shaped modes: samples whose decoded rows are compared with compressor(X)")
    args = ap.parse_args()

    weights = os.environ.get("LOSSYLESS_CLIP_WEIGHTS", "synthetic")
    shaped = args.stl10_shaped or args.imagenet_shaped > 0
    loader = dict(batch_size=args.batch)
    Y = Yt = test = None
    if shaped:
        from workloads import MixedSizeImages, Stl10Shaped
        loader = dict(batch_size=args.batch, num_workers=args.workers)
    elif args.images:
        train = torch.from_numpy(np.load(args.images))
        data = os.path.basename(args.images)
        Y = np.load(args.labels) if args.labels else None
        test = torch.from_numpy(np.load(args.test_images)) if args.test_images else None
        Yt = np.load(args.test_labels) if args.test_labels else None
    else:
        g = torch.Generator().manual_seed(0)
        train = torch.randint(0, 256, (args.n_synthetic, 96, 96, 3), generator=g, dtype=torch.uint8)
        data = f"synthetic uint8 96x96 x{args.n_synthetic} (no --images: assets absent)"

    for name in ("clip_compressor_b01", "clip_compressor_b005", "clip_compressor_b001"):
        comp, transform = getattr(hubconf, name)(device="cuda", clip_weights=weights, gpu_preprocess=shaped)
        if args.stl10_shaped:
            train, test = Stl10Shaped(5000, transform, split_seed=0), Stl10Shaped(8000, transform, split_seed=1)
            Y, Yt = train.labels, test.labels
            data = "STL10-shaped stand-in: 5000 train / 8000 test generated 96x96 images, 10 classes (STL10 absent)"
        elif args.imagenet_shaped:
            train = MixedSizeImages(args.imagenet_shaped, transform)
            Y = train.targets
            data = (f"ImageNet-val-shaped stand-in: {args.imagenet_shaped} generated photos, "
                    f"{len(set(train.shape_of.tolist()))} sizes from 96x128 to 2448x3264 (ImageNet absent)")
        n = len(train)
        with tempfile.TemporaryDirectory() as d:
            f, lf = os.path.join(d, "Z.bin"), (os.path.join(d, "Y.npy") if shaped else None)
            enc = _compress(comp, train, f, lf, loader)
            bits = 8 * os.path.getsize(f) / n
            t0 = time.perf_counter()
            Z = comp.decompress_dataset(f, label_file=lf, is_info=False)
            dec = time.perf_counter() - t0
            if shaped:
                Z, Yfile = Z
                assert np.array_equal(Yfile, np.asarray(Y) % 65536)      # labels ride as uint16 (hub/compressor.py:189)
            assert Z.shape == (n, 512)
            # the file decodes to exactly what compressor(X) returns (all samples, or an evenly spread subset)
            idx = np.arange(n) if (not shaped or n <= args.check) else np.linspace(0, n - 1, args.check).astype(np.int64)
            ref = _represent(comp, train, idx, args.batch)
            assert np.array_equal(Z[idx], ref)
            acc = "skipped: labels / test split not given"
            if Y is not None and test is not None and Yt is not None:
                from sklearn.svm import LinearSVC
                ft = os.path.join(d, "Zt.bin")
                _compress(comp, test, ft, None, loader)
                Zt = comp.decompress_dataset(ft, is_info=False)
                clf = LinearSVC(C=7e-3).fit(Z, Y)          # README.md:75
                acc = float(clf.score(Zt, Yt))
        print(json.dumps(dict(rate_point=name, data=data, clip_weights=weights, images=n,
                              call=("Dataset(transform=RawRGB) -> compress_dataset(dataset, file, label_file, "
                                    f"dict(batch_size={args.batch}, num_workers={args.workers}))") if shaped
                              else "tensor fast path",
                              bits_per_img=round(bits, 2), encode_img_per_sec=round(n / enc, 1),
                              decode_img_per_sec=round(n / dec, 1), round_trip="exact",
                              round_trip_samples=int(len(idx)), linear_svc_accuracy=acc)), flush=True)


if __name__ == "__main__":
    main()
