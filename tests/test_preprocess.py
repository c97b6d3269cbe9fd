"""CLIP preprocessing: the host-side tap tables against real Pillow (CPU), and the HIP kernel
against the PIL transform chain (GPU).  Pillow's 8-bit resampler is integer arithmetic, so
both comparisons are bit-exact."""
import numpy as np
import pytest
import torch
from PIL import Image

from lossyless_amd.preprocess import (ClipPreprocess, crop_origin, pillow_bicubic_taps,
                                      resized_size)

SIZES = [(96, 96), (375, 500), (500, 333), (224, 224), (64, 80), (230, 224), (1000, 700)]  # (H, W)


def _resample_numpy(img, ow, oh):
    """two-pass 22-bit fixed-point resample with a uint8 intermediate, driven by our tables"""
    H, W, _ = img.shape
    hb, hk = pillow_bicubic_taps(W, ow, 0, ow)
    vb, vk = pillow_bicubic_taps(H, oh, 0, oh)
    tmp = np.zeros((H, ow, 3), np.uint8)
    for x in range(ow):
        a, n = hb[x]
        acc = (1 << 21) + (img[:, a:a + n].astype(np.int64) * hk[x, :n][None, :, None]).sum(1)
        tmp[:, x] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((oh, ow, 3), np.uint8)
    for y in range(oh):
        a, n = vb[y]
        acc = (1 << 21) + (tmp[a:a + n].astype(np.int64) * vk[y, :n][:, None, None]).sum(0)
        out[y] = np.clip(acc >> 22, 0, 255)
    return out


@pytest.mark.parametrize("H,W", SIZES[:5])
def test_tap_tables_reproduce_pillow_bytes(H, W):
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ow, oh = resized_size(W, H)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    assert np.array_equal(_resample_numpy(img, ow, oh), ref)


def test_resize_and_crop_geometry():
    assert resized_size(96, 96) == (224, 224)
    assert resized_size(500, 375) == (298, 224)          # int(224 * 500 / 375)
    assert resized_size(333, 500) == (224, 336)
    assert crop_origin(298, 224) == (37, 0)
    assert crop_origin(225, 224) == (0, 0)               # round(0.5) -> 0 (banker's)
    assert crop_origin(227, 224) == (2, 0)               # round(1.5) -> 2
    x = ClipPreprocess()(Image.fromarray(np.zeros((375, 500, 3), np.uint8)))
    assert tuple(x.shape) == (3, 224, 224)
    # black pixel -> (0 - mean) / std
    assert torch.allclose(x[:, 0, 0], torch.tensor([-0.48145466 / 0.26862954, -0.4578275 / 0.26130258,
                                                    -0.40821073 / 0.27577711]))


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", SIZES)
def test_gpu_preprocess_is_bit_identical_to_pil_chain(H, W):
    from lossyless_amd.preprocess import ClipPreprocessGPU
    rng = np.random.default_rng(H + 7 * W)
    imgs = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    # low-contrast + saturated images exercise clip8 on both ends (bicubic overshoot)
    imgs[1] = np.where(rng.random((H, W, 3)) < 0.5, 0, 255)
    pil = ClipPreprocess()
    want = torch.stack([pil(Image.fromarray(im)) for im in imgs]).half().permute(0, 2, 3, 1).contiguous()
    got = ClipPreprocessGPU()(torch.from_numpy(imgs).cuda())
    assert got.shape == (3, 224, 224, 3) and got.dtype == torch.float16
    assert torch.equal(got.cpu(), want)


@pytest.mark.gpu
def test_compressor_accepts_raw_uint8_batches():
    import hubconf
    comp, transform = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (6, 96, 96, 3), dtype=np.uint8)          # STL10-shaped
    x = torch.stack([transform(Image.fromarray(im)) for im in raw]).half()  # reference-style input
    assert comp.compress(torch.from_numpy(raw).cuda()) == comp.compress(x.permute(0, 2, 3, 1).contiguous().cuda())
