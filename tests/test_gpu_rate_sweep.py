"""GPU: the BASELINE.json configs[2] / configs[4] harness (tools/rate_sweep.py: rate sweep over the three
shipped rate points, file -> compressor(X) exact round trip, LinearSVC(C=7e-3) as in the reference's
README.md:75 and notebooks/Hub.ipynb) driven end to end on small synthetic .npy files.  The reference's
numbers (1506.6 bits/img, 98.64 %) need STL10 + the OpenAI ViT-B-32.pt, which cannot be fetched offline;
what is executed here is the harness itself, on data whose class is recoverable from the pixels."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_rate_sweep_harness_end_to_end(tmp_path):
    rng = np.random.default_rng(0)

    def make(n):
        y = rng.integers(0, 4, size=n)
        x = rng.integers(0, 256, size=(n, 96, 96, 3), dtype=np.uint8)
        # class-dependent colour cast + gradient: separable after any reasonable featuriser
        for k in range(4):
            m = y == k
            x[m, :, :, k % 3] = np.clip(x[m, :, :, k % 3].astype(np.int32) // 4 + 150 + 20 * (k // 3), 0, 255).astype(np.uint8)
        return x, y.astype(np.int64)

    Xtr, Ytr = make(256)
    Xte, Yte = make(96)
    paths = {}
    for name, arr in (("X", Xtr), ("Y", Ytr), ("Xt", Xte), ("Yt", Yte)):
        paths[name] = str(tmp_path / f"{name}.npy")
        np.save(paths[name], arr)
    env = dict(os.environ, LOSSYLESS_CLIP_WEIGHTS="synthetic")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rate_sweep.py"), "--images", paths["X"],
                        "--labels", paths["Y"], "--test-images", paths["Xt"], "--test-labels", paths["Yt"],
                        "--batch", "64"], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [x["rate_point"] for x in rows] == ["clip_compressor_b01", "clip_compressor_b005", "clip_compressor_b001"]
    for x in rows:
        assert x["round_trip"] == "exact" and x["images"] == 256 and x["clip_weights"] == "synthetic"
        assert isinstance(x["linear_svc_accuracy"], float) and 0.0 <= x["linear_svc_accuracy"] <= 1.0
        assert x["bits_per_img"] > 64 and x["encode_img_per_sec"] > 0 and x["decode_img_per_sec"] > 0
    # finer quantisation (smaller beta) never costs fewer bits on the same embeddings
    assert rows[0]["bits_per_img"] <= rows[1]["bits_per_img"] <= rows[2]["bits_per_img"]
    # the class signal survives the (random-weight) tower + quantisation well above chance
    assert max(x["linear_svc_accuracy"] for x in rows) > 0.5
