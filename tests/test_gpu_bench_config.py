"""GPU: parity AT THE CONFIGURATION bench.py TIMES (BASELINE.json configs[1]: batch 1024, what
hub/compressor.py:186-187 does per batch) -- the kernel instantiations a 1024-image batch
dispatches to (persistent patch-embed GEMM with the staged EPI_PATCH epilogue, persistent
256/320-row tiles, the class-token-pruned last block with row stride 50*768) are different code
from what small batches run, so they get their own comparisons:

  * lla_patch_embed_f16 (A_PATCH_NHWC / A_PATCH_NCHW + EPI_PATCH) against an fp64 conv2d at
    M >= 9000, and bit-identity with the small-batch kernels;
  * the tower at B = 1024 (both layouts) and ragged large batches (257, 1000): bit-identical to
    the same images pushed through in slices of 8 (the oracle-checked path), <= 1e-3 from
    oracle/vit.py on a 32-image subset;
  * compress_dataset over those batches == the oracle's container for those embeddings.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_tables
from lossyless_amd import _lib
from oracle import cbind, container, eb
from oracle import vit as ovit

pytestmark = pytest.mark.gpu


def synth_images(B, seed=0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).to(device)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=device)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=device)
    return ((u8.float() / 255 - mean) / std).half()


@pytest.fixture(scope="module")
def comp():
    import hubconf
    c, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    return c


@pytest.fixture(scope="module")
def big(comp):
    """1024 NHWC images and their embeddings through the full-batch path."""
    x = synth_images(1024, seed=31)
    return x, comp.clip(x)


def _patch_embed(x, layout, conv, pos):
    B = x.shape[0]
    out = torch.full((B * 50, 768), float("nan"), dtype=torch.float32, device="cuda")
    K_order = conv.permute(0, 2, 3, 1) if layout == _lib.LLA_LAYOUT_NHWC else conv
    w = K_order.reshape(768, -1).half().contiguous().cuda()
    rc = _lib.lib().lla_patch_embed_f16(_lib.ptr(x), layout, B, _lib.ptr(w), _lib.ptr(pos), _lib.ptr(out),
                                        _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    return out.view(B, 50, 768)


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_patch_embed_gemm_against_fp64_conv(layout):
    """B = 200 -> M = 9800 patch rows: the persistent kernel's gather modes and staged EPI_PATCH
    epilogue (ragged last row tile included), vs conv2d evaluated in float64."""
    g = torch.Generator().manual_seed(3)
    conv = torch.randn(768, 3, 32, 32, generator=g) * 0.02
    pos = (torch.randn(50, 768, generator=g) * 0.05).cuda()
    x = synth_images(200, seed=5)                                   # NHWC
    lay = _lib.LLA_LAYOUT_NHWC if layout == "nhwc" else _lib.LLA_LAYOUT_NCHW
    xin = x if layout == "nhwc" else x.permute(0, 3, 1, 2).contiguous()
    got = _patch_embed(xin, lay, conv, pos)
    assert bool(torch.isnan(got[:, 0]).all())                       # class rows are not written
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), conv.half().double().cuda(),
                                     stride=32)                     # [B,768,7,7]
    ref = ref.flatten(2).transpose(1, 2) + pos[1:].double()
    err = (got[:, 1:].double() - ref).abs()
    assert float(err.max()) < 2e-4, float(err.max())                # fp32 accumulation of 3072 fp16 products
    # the same rows through the small-batch kernels (M = 392 < 9000), bit for bit
    small = torch.cat([_patch_embed(xin[i:i + 8], lay, conv, pos) for i in range(0, 200, 8)])
    assert torch.equal(got[:, 1:], small[:, 1:])


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_tower_at_batch_1024_equals_slices_of_8_and_the_oracle(comp, big, layout):
    x, z_nhwc = big
    if layout == "nchw":
        xin = x.permute(0, 3, 1, 2).contiguous()
        z = comp.clip(xin)      # (K runs (c,kh,kw) instead of (kh,kw,c): close to, not bitwise, the NHWC result)
        assert float((z.float() - z_nhwc.float()).norm() / z_nhwc.float().norm()) < 1e-3
    else:
        xin, z = x, z_nhwc
    z8 = torch.cat([comp.clip(xin[i:i + 8]) for i in range(0, 1024, 8)])
    assert torch.equal(z, z8), "batch-1024 kernels differ bitwise from the small-batch kernels"
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    idx = torch.arange(0, 1024, 32)                                  # 32 images spread over the batch
    ref = ovit.vit_b32_forward(synthetic_vit_state_dict(1),
                               x[idx].permute(0, 3, 1, 2).float().cpu()).numpy()
    got = z[idx].float().cpu().numpy()
    rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 1e-3, rel.max()


@pytest.mark.parametrize("B", [257, 1000])
def test_tower_ragged_large_batches(comp, big, B):
    x, z = big
    zb = comp.clip(x[:B].contiguous())
    assert torch.equal(zb, z[:B])                                    # per-image results do not depend on the batch
    xc = x[:B].permute(0, 3, 1, 2).contiguous()
    zc = comp.clip(xc)
    assert torch.equal(zc[:8], comp.clip(xc[:8].contiguous()))
    assert float((zc.float() - z[:B].float()).norm() / z[:B].float().norm()) < 1e-3


def test_compress_dataset_at_batch_1024_is_the_oracle_container(comp, big, tmp_path):
    x, z = big
    tab = load_tables("5e-02")
    sym = eb.symbols_of(z.float().cpu().numpy(), tab)
    pay, off = cbind.rans_encode_batch(sym, tab["cdf"], tab["cdf_len"], tab["offset"])
    pay = pay.tobytes()
    want = container.container_bytes([pay[int(off[i]):int(off[i + 1])] for i in range(len(sym))])
    for bs, name in ((1024, "a"), (257, "b")):                       # one full batch / ragged batches
        f = tmp_path / f"{name}.bin"
        comp.compress_dataset(x, f, kwargs_dataloader=dict(batch_size=bs), is_info=False)
        assert f.read_bytes() == want
    Z = comp.decompress_dataset(f, is_info=False, is_cpu=False)
    assert np.array_equal(Z, eb.dequantise(sym, tab))
    assert np.array_equal(Z, comp.decompress_dataset(f, is_info=False, is_cpu=True))   # host decoder
    assert np.array_equal(Z, comp(x).cpu().numpy())


def test_bench_verifier_reports_true(comp, big):
    import bench
    x, _ = big
    v = bench.verify_first_batch(comp, x)
    assert v["records_equal_oracle"] is True and v["embedding_ok"] is True and v["images"] == 1024
