"""GPU: direct oracle parity AT THE TIMED PASS SIZE, and the table rows under the GPU run's eye.

  * ONE tower pass of 8704 images -- what `RecordStream` hands the library per call in bench.py and
    `compress_dataset` (hub/compressor.py:93 `self.clip(X)` at the size it really runs here: nine pushed device
    batches read in place through `lla_vit_b32_forward_gather`, the four-wave GEMMs at M = 435 200) -- with 64 images
    spread over the pass checked against oracle/vit.py (fp32 CPU tower, <= 1e-3 relative L2, the north_star
    tolerance), and the WHOLE container of the pass checked byte for byte against the oracle coder fed the same
    embeddings (hub/compressor.py:98, 192-196).
  * SURVEY.md rows A11 / A12 (`EntropyBottleneck.update()`, `pmf_to_quantized_cdf`; hub/compressor.py:63): the
    CPU-marked tests of tests/test_host.py again under `-m gpu`, so that a regression in table derivation shows
    in the driver's GPU record too (the device tables the coder reads are compared with the golden fixtures).
"""
import numpy as np
import pytest
import torch

from conftest import BETAS, load_tables
from oracle import cbind, container, eb
from oracle import vit as ovit

pytestmark = pytest.mark.gpu

PASS = 8704


def _images(B, seed):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).cuda()
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device="cuda")
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device="cuda")
    return ((u8.float() / 255 - mean) / std).half()


def test_one_timed_size_pass_against_the_oracle_tower_and_the_oracle_coder():
    import hubconf
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    from lossyless_amd.compressor import _TOWER_BATCH
    assert _TOWER_BATCH == PASS
    comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    batches = [_images(1024, 700 + i) for i in range(9)]
    batches[-1] = batches[-1][:PASS - 8 * 1024].contiguous()           # 8 x 1024 + 512 = one pass exactly
    stream = comp.record_stream(16)
    for b in batches:
        stream.push(b, donate=True)
    body = stream.finish()
    assert stream.gathered_passes >= 1, "the pass did not take the in-place (gather) path"
    x = torch.cat(batches)
    assert x.shape[0] == PASS
    # the embeddings of the same pass (same kernels: one library call of 8704 images)
    z = comp.clip(x)
    # (1) 64 images spread over the pass vs the fp32 CPU tower
    idx = torch.arange(0, PASS, PASS // 64)[:64]
    ref = ovit.vit_b32_forward(synthetic_vit_state_dict(1), x[idx].permute(0, 3, 1, 2).float().cpu()).numpy()
    got = z[idx].float().cpu().numpy()
    rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 1e-3, rel.max()
    # (2) the whole container of the pass == the oracle's coding of the same embeddings
    tab = load_tables("5e-02")
    sym = eb.symbols_of(z.float().cpu().numpy(), tab)
    pay, off = cbind.rans_encode_batch(sym, tab["cdf"], tab["cdf_len"], tab["offset"])
    pay = pay.tobytes()
    want = container.container_bytes([pay[int(off[i]):int(off[i + 1])] for i in range(PASS)])
    assert bytes(want[:4]) == PASS.to_bytes(4, "big")
    assert body.tobytes() == want[4:]


@pytest.mark.parametrize("tag", BETAS)
def test_update_reproduces_frozen_tables_and_the_device_tables_are_the_golden_ones(tag):
    """A11 under `-m gpu`: the hub module's tables ON THE DEVICE (what the coder kernels read) are the committed
    integer tables, `update()` is a no-op on them, and `update(force=True)` re-derives them (fp32 torch-CPU + the
    C-ABI `lla_pmf_to_quantized_cdf`) to the same integers up to libm-level edge moves."""
    import hubconf
    name = {"1e-01": "clip_compressor_b01", "5e-02": "clip_compressor_b005", "1e-02": "clip_compressor_b001"}[tag]
    comp, _ = getattr(hubconf, name)(device="cuda", clip_weights="synthetic")
    tab = load_tables(tag)
    t = comp._tables()
    assert t["cdf"].is_cuda and np.array_equal(t["cdf"].cpu().numpy(), tab["cdf"])
    assert np.array_equal(t["cdf_len"].cpu().numpy(), tab["cdf_len"])
    assert np.array_equal(t["offset"].cpu().numpy(), tab["offset"])
    assert np.array_equal(t["exp_scale"].cpu().numpy(), tab["exp_scale"])
    m = comp.entropy_bottleneck
    assert m.update() is False
    assert m.update(force=True) is True
    diff = np.abs(m._quantized_cdf.cpu().numpy().astype(np.int64) - tab["cdf"])
    assert (diff != 0).mean() < 0.02
    assert np.array_equal(m._cdf_length.cpu().numpy(), tab["cdf_len"])
    assert np.array_equal(m._offset.cpu().numpy(), tab["offset"])


def test_pmf_to_quantized_cdf_matches_oracle_under_the_gpu_run():
    """A12 under `-m gpu` (same cases as tests/test_host.py): the library's host entry point against the oracle's C."""
    from lossyless_amd.entropy import pmf_to_quantized_cdf
    rng = np.random.default_rng(0)
    checked = 0
    for n in (1, 2, 5, 31, 32):
        for _ in range(50):
            p = rng.dirichlet(np.full(n, 0.3)).astype(np.float32)
            p[rng.random(n) < 0.2] *= 1e-7
            if p.sum() <= 0:
                continue
            try:
                want = cbind.pmf_to_quantized_cdf(p)
            except ValueError:
                with pytest.raises(RuntimeError):
                    pmf_to_quantized_cdf(p)
                continue
            assert np.array_equal(pmf_to_quantized_cdf(p), want)
            checked += 1
    assert checked > 150
