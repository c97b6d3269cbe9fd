"""GPU: BASELINE.json configs[2], [3], [4] at their real SCALE and SHAPES, on generated stand-ins for the data
(ImageNet-val, STL10 and the OpenAI ViT-B-32.pt cannot be fetched offline; what stays asset-gated is only the
comparison with the reference's recorded numbers -- 1506.6 bits/img, 98.64 %).

  configs[3]  1 000 000 lazily generated 224x224 images through `compress_dataset`, 1 rank vs 2 ranks (gloo, both
              on the one GPU of the test box; the 8-GPU RCCL run is the driver's): same file, N = 10^6 in the
              header, a 65 536-record slice decodes to exactly `compressor(X)` of the same images; and an 8-rank
              run with ragged and EMPTY shards (N < world).
  configs[2]  50 000 photos of the ImageNet-val size mix through the three rate points by the unchanged reference
              call with `gpu_preprocess=True`: labels intact, exact round trip, monotone bits/img.
  configs[4]  STL10's real split sizes (5 000 / 8 000, 96x96) through the reference call, decompress_dataset,
              LinearSVC(C=7e-3).
"""
import hashlib
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LOSSYLESS_CLIP_WEIGHTS="synthetic")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _bench(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", *args],
                       env=_env(), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def test_config3_one_million_images_one_rank_vs_two(tmp_path):
    """the file of a sharded run IS the unsharded file -- asserted by sha, no retry, no tolerance:
    (a) two ranks (two GPUs under nccl when the box has them; on a 1-GPU box two gloo ranks = two PROCESSES on the GPU,
    which `bench.py` then gives disjoint contiguous halves of the CU mask: two processes side by side on the same CUs' part
    of the chip is the one configuration in which a record in 10^2 .. 10^7 images comes out a quantisation step off --
    DESIGN.md 5.4: 24 of 24 runs differ without a mask, 0 of 82 with the halves, one process is always right); (b) the two shards computed one after the
    other by ONE process and concatenated.  A mismatch fails with tools/diff_containers.py's classification."""
    n = 1_000_000
    one, two = str(tmp_path / "one.bin"), str(tmp_path / "two.bin")
    r1 = _bench("--gpus", "1", "--dataset-images", str(n), "--keep-file", one)
    shared_gpu = torch.cuda.device_count() < 2
    r2 = _bench("--gpus", "2", "--backend", "gloo" if shared_gpu else "nccl", "--dataset-images", str(n), "--keep-file", two)
    assert r1["images"] == r2["images"] == n and r2["n_gpus"] == 2 and r2["comm"]["world_size"] == 2
    if shared_gpu:
        assert r2["comm"]["cu_mask"] == "0:0..127 (128 CUs)", r2["comm"]      # rank 0's half of the CU mask
    if r1["file_sha256"] != r2["file_sha256"]:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from diff_containers import diff_containers
        pytest.fail("1-rank and 2-rank files differ: " + json.dumps(diff_containers(one, two, ranks=2)))
    assert r1["file_sha256"] == _sha(one) == _sha(two)

    # (b) the two shards by ONE process, one after the other, concatenated: strictly the unsharded file
    import hubconf
    from lossyless_amd.compressor import SyntheticImages
    from lossyless_amd.distributed import shard_bounds
    comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    ds, h = SyntheticImages(n), hashlib.sha256()
    h.update(struct.pack(">I", n))
    for rank in range(2):
        lo, hi = shard_bounds(n, rank, 2)
        stream = comp.record_stream()
        for a in range(lo, hi, 8704):
            stream.push(ds.device_batch(a, min(a + 8704, hi), "cuda"), donate=True)
            if (a - lo) // 8704 % 16 == 15:
                h.update(stream.finish().tobytes())
        h.update(stream.finish().tobytes())
    assert h.hexdigest() == r1["file_sha256"], "the concatenation of the two shards is not the unsharded file"
    del comp
    assert r1["value"] > 30e3, r1         # tower-bound, not generator-bound (66k in round 2 with the torch generator)
    with open(one, "rb") as f:
        assert struct.unpack(">I", f.read(4))[0] == n
    os.remove(two)

    # a 65 536-record slice of the file == compressor(X) on the same images, on both decoders
    import ctypes
    import hubconf
    from lossyless_amd import _lib
    from lossyless_amd.compressor import SyntheticImages
    blob = np.fromfile(one, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    cnt = ctypes.c_uint32(0)
    _lib.check(_lib.lib().lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                              off.ctypes.data_as(ctypes.c_void_p), off.size, ctypes.byref(cnt)),
               "lla_container_index")
    assert cnt.value == n and int(off[n]) + 4 == blob.size
    lo, m = 300_000, 65_536
    sub = str(tmp_path / "slice.bin")
    with open(sub, "wb") as f:
        f.write(struct.pack(">I", m))
        f.write(blob[4 + int(off[lo]):4 + int(off[lo + m])].tobytes())
    del blob
    comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    z_gpu = comp.decompress_dataset(sub, is_info=False, is_cpu=False)
    z_cpu = comp.decompress_dataset(sub, is_info=False)                 # the reference's default: host coder
    assert z_gpu.shape == (m, 512) and np.array_equal(z_gpu, z_cpu)
    ds = SyntheticImages(n)
    for i in range(0, m, 8192):
        want = torch.cat([comp(ds.device_batch(lo + j, lo + j + 1024, "cuda")) for j in range(i, i + 8192, 1024)])
        assert np.array_equal(z_gpu[i:i + 8192], want.cpu().numpy()), i


def test_config3_eight_ranks_ragged_and_empty_shards():
    """World size 8 (gloo, all ranks on the one GPU): 4099 images = shards of 513 / 512, and 5 images = three
    EMPTY shards; rank 0's file equals the 1-rank file."""
    for n in (4099, 5):
        r8 = _bench("--gpus", "8", "--backend", "gloo", "--batch", "256", "--dataset-images", str(n))
        r1 = _bench("--gpus", "1", "--batch", "256", "--dataset-images", str(n))
        assert r8["n_gpus"] == 8 and r8["images"] == n and r8["comm"]["world_size"] == 8
        assert r8["comm"]["torch_threads"] >= 1 and r8["comm"]["host_cpus_per_rank"] >= 1
        assert r8["file_sha256"] == r1["file_sha256"] and r8["bits_per_img"] == r1["bits_per_img"]


def _sweep(*args, timeout=1100):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rate_sweep.py"), *args], env=_env(),
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert [x["rate_point"] for x in rows] == ["clip_compressor_b01", "clip_compressor_b005", "clip_compressor_b001"]
    return rows


def test_config2_imagenet_val_shaped_rate_sweep():
    rows = _sweep("--imagenet-shaped", "50000", "--batch", "256", "--workers", "16", "--check", "2048")
    for x in rows:
        assert x["images"] == 50000 and x["round_trip"] == "exact" and x["round_trip_samples"] == 2048
        assert "compress_dataset(dataset" in x["call"] and x["encode_img_per_sec"] > 0
    assert rows[0]["bits_per_img"] <= rows[1]["bits_per_img"] <= rows[2]["bits_per_img"]


def test_config4_stl10_shaped_round_trip_and_linear_svc():
    rows = _sweep("--stl10-shaped", "--batch", "128", "--workers", "16", "--check", "5000")
    for x in rows:
        assert x["images"] == 5000 and x["round_trip"] == "exact" and x["round_trip_samples"] == 5000
        assert isinstance(x["linear_svc_accuracy"], float)
    assert rows[0]["bits_per_img"] <= rows[1]["bits_per_img"] <= rows[2]["bits_per_img"]
    # 10 classes, chance = 0.1: the class signal survives the (random-weight) tower and every quantiser
    assert min(x["linear_svc_accuracy"] for x in rows) > 0.5


@pytest.mark.slow
def test_soak_six_million_images_twice_give_the_same_records():
    """Determinism soak (docs/history/DESIGN_rounds_1-5.md 5.3): 6 M lazily generated images through the streaming encoder,
    twice -- the SHA-256 of all records must be equal.  Round 3's two-lane tower failed this kind of run at one
    embedding per 10^6..10^8 images; the product build runs one stream (0 events in 39 M images then).  ~2 x 65 s:
    under `-m "gpu and slow"` since round 6 (VERDICT r5 #4b).  The plain `-m gpu` run keeps a 10^6-image version of the
    same check inside test_config3_one_million_images_one_rank_vs_two: the file of a `bench.py` process and the records
    of a second pass over the same 10^6 images made in THIS process must have the same sha."""
    import hubconf
    from lossyless_amd.compressor import SyntheticImages
    comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    n, step = 6_000_000, 8704
    ds = SyntheticImages(n, seed=3)
    digests, counts = [], []
    for _ in range(2):
        h, stream, nbytes = hashlib.sha256(), comp.record_stream(), 0
        for k, lo in enumerate(range(0, n, step)):
            stream.push(ds.device_batch(lo, min(lo + step, n), "cuda"), donate=True)
            if k % 128 == 127:                       # keep the host copy of the records small
                body = stream.finish()
                h.update(body.tobytes())
                nbytes += body.size
        body = stream.finish()
        h.update(body.tobytes())
        digests.append(h.hexdigest())
        counts.append(nbytes + body.size)
    assert counts[0] == counts[1] and counts[0] > 100 * n
    assert digests[0] == digests[1], "two runs over the same 6 M images wrote different records"
