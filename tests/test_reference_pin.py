"""The live pin of the oracle, wired to fire the moment it can.

The arithmetic of this path lives in ``compressai==1.1.5`` and ``clip==1.0``
(/root/reference/hub/compressor.py:5,12-13,39; requirements/environment.yaml:98,105), neither vendored in the
reference nor importable in the build container: parity is "unpinned" until one of these tests RUNS.  In any
environment that gains the package (or the CLIP weights) they stop skipping and turn the claim green or red
without a human: nothing here needs editing.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_oracle_and_fixtures_against_the_real_compressai():
    """tools/verify_against_compressai.py: real ``EntropyBottleneck.update()`` tables vs tests/golden/tables_*,
    real ``pmf_to_quantized_cdf`` on the KATs + 500 random pmfs vs the oracle and the C-ABI, real ``RansEncoder``
    on the golden symbols vs tests/golden/golden_*.bin, real ``RansDecoder`` back."""
    pytest.importorskip("compressai", reason="compressai==1.1.5 is not installed here (no network): the oracle "
                        "stays pinned only by KATs / independent restatements -- parity unpinned")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_against_compressai.py")],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode != 77, "compressai imported in pytest but not in the tool:\n" + r.stdout
    assert r.returncode == 0, "the oracle / fixtures DIFFER from compressai:\n" + r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_recorded_rate_of_the_reference_on_real_clip_weights():
    """``notebooks/Hub.ipynb:253,267,415``: 1506.62 bits/img on STL10 train at beta = 5e-2, 98.64 % LinearSVC accuracy.
    Needs the OpenAI ViT-B/32 weights (``$LOSSYLESS_CLIP_WEIGHTS``) and STL10 as uint8 arrays
    (``$LOSSYLESS_STL10_TRAIN_X / _TRAIN_Y / _TEST_X / _TEST_Y``: .npy files, [N,96,96,3] and [N])."""
    w = os.environ.get("LOSSYLESS_CLIP_WEIGHTS")
    names = ("LOSSYLESS_STL10_TRAIN_X", "LOSSYLESS_STL10_TRAIN_Y", "LOSSYLESS_STL10_TEST_X", "LOSSYLESS_STL10_TEST_Y")
    paths = [os.environ.get(n) for n in names]
    if not w or not os.path.exists(w):
        pytest.skip("real CLIP ViT-B/32 weights absent ($LOSSYLESS_CLIP_WEIGHTS): the reference's 1506.62 bits/img "
                    "/ 98.64 % cannot be reproduced offline")
    if not all(p and os.path.exists(p) for p in paths):
        pytest.skip("STL10 arrays absent ($LOSSYLESS_STL10_TRAIN_X ...): the recorded rate needs the real images")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rate_sweep.py"), "--images", paths[0],
                        "--labels", paths[1], "--test-images", paths[2], "--test-labels", paths[3]],
                       capture_output=True, text=True, timeout=3600, env=dict(os.environ))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    b005 = next(x for x in rows if "005" in str(x["rate_point"]) or "5e-02" in str(x["rate_point"]))
    # the fp16 tower is not the reference's fp16 tower bit for bit (a symbol flips where z sits on a .5 boundary):
    # 0.5 % on the rate, 0.5 points on the accuracy
    assert abs(b005["bits_per_img"] - 1506.62) < 0.005 * 1506.62, b005
    assert abs(100 * float(b005["linear_svc_accuracy"]) - 98.64) < 0.5, b005
