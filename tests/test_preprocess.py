"""CLIP preprocessing: the host-side tap tables against real Pillow (CPU), and the HIP kernel
against the PIL transform chain (GPU).  Pillow's 8-bit resampler is integer arithmetic, so
both comparisons are bit-exact."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import ROOT

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


@pytest.mark.gpu
def test_non_rgb_images_through_the_gpu_chain_equal_the_pil_chain():
    """A CMYK / RGBA / palette image handed over by RawRGB (the finished 224 x 224 crop) next to ordinary RGB photos in one
    ragged batch: every row of the GPU chain's output is the PIL chain's."""
    from lossyless_amd.preprocess import ClipPreprocessGPU, RaggedImages, RawRGB
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (300, 260, 4), dtype=np.uint8)
    pil = [Image.fromarray(a, "CMYK"), Image.fromarray(a[:, :, :3], "RGB"), Image.fromarray(a, "RGBA"),
           Image.fromarray(a[:, :, :3], "RGB").quantize(32), Image.fromarray(a[:, :, 0], "L")]
    raw = [RawRGB()(im) for im in pil]
    got = ClipPreprocessGPU()(RaggedImages.from_list(raw).to("cuda"))
    want = torch.stack([ClipPreprocess()(im).half().permute(1, 2, 0).contiguous() for im in pil])
    assert torch.equal(got.cpu(), want)


# ---------------------------------------------------------------------------------------------------------
# ragged batches (BASELINE configs[2]: ImageNet-val photos of every size) and the drop-in raw transform
# ---------------------------------------------------------------------------------------------------------
# (H, W): the common ImageNet-val shapes, STL10, the identity size, odd aspect ratios, tiny and huge images
RAGGED = [(375, 500), (500, 375), (333, 500), (500, 500), (96, 96), (224, 224), (480, 640), (225, 223),
          (57, 1001), (31, 40), (1200, 1600), (2448, 3264)]


def _pil_chain_nhwc(img):
    return ClipPreprocess()(Image.fromarray(img)).half().permute(1, 2, 0).contiguous()


@pytest.mark.parametrize("H,W", [(375, 500), (500, 333), (1200, 1600), (225, 223)])
def test_cropped_tap_tables_reproduce_the_pil_chain(H, W):
    """Down-scaling windows (7-15 taps) of the 224 cropped columns / rows: resize -> centre crop by PIL ==
    integer two-pass resample driven by ``lla_pillow_bicubic_taps``."""
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    nw, nh = resized_size(W, H)
    left, top = crop_origin(nw, nh)
    hb, hk = pillow_bicubic_taps(W, nw, left, 224)
    vb, vk = pillow_bicubic_taps(H, nh, top, 224)
    tmp = np.zeros((H, 224, 3), np.int64)
    for x in range(224):
        a, n = hb[x]
        tmp[:, x] = np.clip(((1 << 21) + (img[:, a:a + n].astype(np.int64) * hk[x, :n][None, :, None]).sum(1)) >> 22, 0, 255)
    out = np.zeros((224, 224, 3), np.int64)
    for y in range(224):
        a, n = vb[y]
        out[y] = np.clip(((1 << 21) + (tmp[a:a + n] * vk[y, :n][:, None, None]).sum(0)) >> 22, 0, 255)
    want = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC).crop((left, top, left + 224, top + 224)))
    assert np.array_equal(out.astype(np.uint8), want)


def test_raw_transform_and_ragged_collate_on_cpu():
    from lossyless_amd.preprocess import RaggedImages, RawRGB, ragged_collate
    rng = np.random.default_rng(0)
    t = RawRGB()
    a = rng.integers(0, 256, (40, 30, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (20, 50, 3), dtype=np.uint8)
    ta, tb = t(Image.fromarray(a)), t(b)
    assert ta.dtype == torch.uint8 and tuple(ta.shape) == (40, 30, 3) and np.array_equal(ta.numpy(), a)
    assert np.array_equal(t(Image.fromarray(a[:, :, 0])).numpy(), np.repeat(a[:, :, :1], 3, axis=2))   # L -> RGB
    with pytest.raises(ValueError):
        t(torch.zeros(3, 8, 8))
    # equal sizes stack; different sizes become one blob + shapes; labels collate as usual
    x, y = ragged_collate([(ta, 1), (ta, 2)])
    assert isinstance(x, torch.Tensor) and tuple(x.shape) == (2, 40, 30, 3) and y.tolist() == [1, 2]
    x, y = ragged_collate([(ta, 1), (tb, 2), (ta, 3)])
    assert isinstance(x, RaggedImages) and len(x) == 3 and y.tolist() == [1, 2, 3]
    assert x.shapes.tolist() == [[40, 30], [20, 50], [40, 30]] and x.offsets.tolist() == [0, 3600, 6600]
    assert x.blob.numel() == 3600 + 3000 + 3600 + 4
    assert np.array_equal(x.image(1).numpy(), b) and np.array_equal(x.image(2).numpy(), a)
    # float samples (the PIL transform) fall through to the default collate
    x, y = ragged_collate([(torch.zeros(3, 4, 4), 0), (torch.ones(3, 4, 4), 1)])
    assert tuple(x.shape) == (2, 3, 4, 4)


@pytest.mark.parametrize("mode", ["RGBA", "P", "CMYK", "LA"])
def test_non_rgb_images_are_converted_after_resize_and_crop_as_the_reference_does(mode):
    """CLIP's transform is Resize -> CenterCrop -> convert("RGB") (clip.py ``_transform``); for RGBA (resized
    premultiplied), P (nearest-neighbour), CMYK (four channels) converting FIRST gives other pixels.  Both in-repo
    chains follow the reference's order: the host chain, and RawRGB -- which hands such an image over as the finished
    224 x 224 RGB crop, on which the GPU chain's resize and crop are the identity (scale-1 bicubic taps are (0, 1, 0, 0):
    ``test_same_size_taps_are_the_identity``)."""
    from lossyless_amd.preprocess import CLIP_MEAN, CLIP_STD, RawRGB
    rng = np.random.default_rng(5)
    rgba = rng.integers(0, 256, (130, 97, 4), dtype=np.uint8)
    src = {"RGBA": Image.fromarray(rgba, "RGBA"), "CMYK": Image.fromarray(rgba, "CMYK"),
           "LA": Image.fromarray(rgba[:, :, :2], "LA"),
           "P": Image.fromarray(rgba[:, :, :3], "RGB").quantize(16)}[mode]
    assert src.mode == mode
    nw, nh = resized_size(97, 130)
    left, top = crop_origin(nw, nh)
    want = np.asarray(src.resize((nw, nh), Image.BICUBIC).crop((left, top, left + 224, top + 224)).convert("RGB"))
    early = np.asarray(src.convert("RGB").resize((nw, nh), Image.BICUBIC).crop((left, top, left + 224, top + 224)))
    raw = RawRGB()(src)
    assert tuple(raw.shape) == (224, 224, 3) and np.array_equal(raw.numpy(), want)
    t = ClipPreprocess()(src)
    ref = (torch.from_numpy(want.copy()).permute(2, 0, 1).float().div(255)
           - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)
    assert torch.equal(t, ref)
    if mode != "LA":
        assert not np.array_equal(want, early)      # the order is not a formality for these modes


def test_same_size_taps_are_the_identity():
    b, k = pillow_bicubic_taps(224, 224, 0, 224)
    for x in range(224):
        a, n = b[x]
        w = np.zeros(224 + 8, np.int64)
        w[a + 4:a + n + 4] = k[x, :n]
        assert w[x + 4] == 1 << 22 and w.sum() == 1 << 22


class _TwoSizes(torch.utils.data.Dataset):
    def __init__(self, n, ragged):
        self.n, self.ragged = n, ragged

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        h, w = ((8, 6) if i % 2 else (5, 9)) if self.ragged else (7, 7)
        return torch.full((h, w, 3), i % 251, dtype=torch.uint8), i


@pytest.mark.parametrize("ragged", [False, True])
def test_ragged_collate_in_loader_workers_hands_over_shared_memory(ragged):
    """In a DataLoader worker the collated batch is built in shared memory (no second copy on the way to the main
    process); same batches as with ``num_workers=0``."""
    from torch.utils.data import DataLoader
    from lossyless_amd.preprocess import RaggedImages, ragged_collate
    ds = _TwoSizes(23, ragged)
    got = {w: list(DataLoader(ds, batch_size=5, num_workers=w, collate_fn=ragged_collate)) for w in (0, 2)}
    assert len(got[0]) == len(got[2]) == 5
    for (x0, y0), (x2, y2) in zip(got[0], got[2]):
        assert torch.equal(y0, y2)
        if ragged:
            assert isinstance(x2, RaggedImages) and torch.equal(x0.blob, x2.blob)
            assert np.array_equal(x0.shapes, x2.shapes) and np.array_equal(x0.offsets, x2.offsets)
            assert x2.blob.is_shared() and not x0.blob.is_shared()
        else:
            assert torch.equal(x0, x2) and x2.is_shared() and not x0.is_shared()


@pytest.mark.gpu
def test_ragged_gpu_preprocess_is_bit_identical_to_pil_chain():
    """One launch over images of 12 different sizes (down- and up-scaling, identity, a photo too large for a
    one-row band in LDS -> two-pass path): every output equals the per-image PIL chain byte for byte."""
    from lossyless_amd.preprocess import ClipPreprocessGPU, RaggedImages
    rng = np.random.default_rng(11)
    imgs = []
    for k, (H, W) in enumerate(RAGGED * 2):
        im = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        if k % 5 == 1:
            im = np.where(rng.random((H, W, 3)) < 0.5, 0, 255).astype(np.uint8)   # overshoot on both ends
        imgs.append(im)
    want = torch.stack([_pil_chain_nhwc(im) for im in imgs])
    pre = ClipPreprocessGPU()
    rag = RaggedImages.from_list([torch.from_numpy(im) for im in imgs]).to("cuda")
    got = pre(rag)
    assert got.shape == (len(imgs), 224, 224, 3) and got.dtype == torch.float16
    assert torch.equal(got.cpu(), want)
    # a list of device tensors is the same thing; a uniform batch through the ragged entry agrees with the
    # uniform kernel
    small = [torch.from_numpy(im).cuda() for im in imgs[:6]]
    assert torch.equal(pre(small).cpu(), want[:6])
    uni = torch.from_numpy(rng.integers(0, 256, (5, 96, 96, 3), dtype=np.uint8))
    assert torch.equal(pre(RaggedImages.from_list(list(uni)).to("cuda")), pre(uni.cuda()))


class _FolderLike(torch.utils.data.Dataset):
    """torchvision-style dataset (STL10 / ImageFolder): holds decoded images, __getitem__ returns
    (transform(PIL image), target)."""

    def __init__(self, arrays, targets, transform):
        self.arrays, self.targets, self.transform = arrays, targets, transform

    def __len__(self):
        return len(self.arrays)

    def __getitem__(self, i):
        return self.transform(Image.fromarray(self.arrays[i])), self.targets[i]


@pytest.mark.gpu
@pytest.mark.parametrize("uniform", [True, False])
def test_unchanged_reference_call_with_gpu_preprocess_writes_the_same_file(tmp_path, uniform):
    """The README call -- Dataset(transform=transform) -> compress_dataset(dataset, file, label_file,
    kwargs_dataloader) (hub/compressor.py:150-207) -- with ``gpu_preprocess=True``: same .bin, same labels as
    with the PIL transform, for an STL10-shaped dataset and for a mixed-size (ImageNet-shaped) one."""
    import hubconf
    rng = np.random.default_rng(3)
    shapes = [(96, 96)] * 37 if uniform else [RAGGED[i % 10] for i in range(37)]
    arrays = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    targets = [int(t) for t in rng.integers(0, 10, len(arrays))]
    files = []
    for gpu_pre, workers in ((False, 0), (True, 0), (True, 2)):
        comp, transform = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic",
                                                       gpu_preprocess=gpu_pre)
        if not gpu_pre:
            # the PIL chain yields CHW, the GPU chain NHWC; the tower's patch-embedding GEMM walks K in the
            # layout's order, so the two layouts agree to ~1e-6, not bit for bit: hand the PIL pixels over as
            # HWC too, and the files must be IDENTICAL
            pil = transform
            transform = lambda im: pil(im).permute(1, 2, 0)
        ds = _FolderLike(arrays, targets, transform)
        f, lf = tmp_path / f"z{gpu_pre}{workers}.bin", tmp_path / f"y{gpu_pre}{workers}.npy"
        comp.compress_dataset(ds, str(f), label_file=str(lf),
                              kwargs_dataloader=dict(batch_size=16, num_workers=workers), is_info=False)
        files.append((f.read_bytes(), np.load(lf)))
    assert files[0][0] == files[1][0] == files[2][0]
    assert np.array_equal(files[0][1], files[1][1]) and np.array_equal(files[0][1], files[2][1])
    assert files[0][1].tolist() == targets
    # compressor(X) / compress(X) take the raw images directly
    comp, transform = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=True)
    raw = [transform(Image.fromarray(a)) for a in arrays[:5]]
    x = torch.stack([_pil_chain_nhwc(a) for a in arrays[:5]]).cuda()
    assert comp.compress(raw) == comp.compress(x)


def _vm_flags_of(addr):
    """VmFlags of the mapping that holds ``addr`` (/proc/self/smaps)."""
    inside = False
    with open("/proc/self/smaps") as f:
        for line in f:
            head = line.split()[0] if line.strip() else ""
            if "-" in head and not line.startswith("VmFlags"):
                try:
                    lo, hi = (int(v, 16) for v in head.split("-"))
                    inside = lo <= addr < hi
                except ValueError:
                    pass
            elif inside and line.startswith("VmFlags:"):
                return line.split()[1:]
    return None


@pytest.mark.gpu
def test_pinned_staging_is_kept_out_of_forked_loader_workers(tmp_path):
    """Userptr-backed pinned pages are write-protected by every fork(); the driver then evicts the GPU queues and
    re-pins (a 16-worker call went 0.7 s -> 11.6 s with one pinned GiB).  The 120 MB - 1.3 GB staging buffers of
    ``_prefetch`` are MADV_DONTFORK ("dc" in the mapping's VmFlags) while the compressor holds them, an ordinary
    block again when it lets go, and a worker-fed call after a tensor call still writes the same file.  (With
    GTT-backed pinned memory -- what importing the package before the runtime starts selects -- every pinned
    mapping is "dc" by nature.)"""
    import copy
    import hubconf
    comp, transform = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=True)
    rng = np.random.default_rng(11)
    raw = torch.from_numpy(rng.integers(0, 256, (300, 96, 96, 3), dtype=np.uint8))
    f0, f1 = tmp_path / "a.bin", tmp_path / "b.bin"
    comp.compress_dataset(raw, str(f0), is_info=False)                 # host tensor -> staged in pinned memory
    bufs = [b for b in comp._staging.buf if b is not None]
    assert bufs and all(b.is_pinned() for b in bufs)
    for b in bufs:
        assert "dc" in _vm_flags_of(b.data_ptr())
    ds = _FolderLike([a.numpy() for a in raw], [0] * len(raw), transform)
    comp.compress_dataset(ds, str(f1), kwargs_dataloader=dict(batch_size=64, num_workers=4), is_info=False)
    assert f0.read_bytes() == f1.read_bytes()
    assert all(b is None for b in copy.deepcopy(comp._staging).buf)   # a copied / pickled one starts with none
    addr = bufs[0].data_ptr()
    keep = bufs[0]
    comp._staging.get(0, keep.numel() * 2, keep.dtype)                 # grows: the old block is released as an ordinary one
    if "dc" not in _vm_flags_of(torch.empty(1 << 20, dtype=torch.uint8).pin_memory().data_ptr()):   # userptr mode
        assert "dc" not in _vm_flags_of(addr)


@pytest.mark.gpu
def test_array_backed_datasets_are_read_from_their_array_and_write_the_same_file(tmp_path):
    """the unchanged reference call on an in-memory dataset (torchvision's STL10 keeps uint8
    [N,3,96,96] in ``.data``) with NO loader arguments skips the per-image Python round trip and the loader workers
    -- ``_array_backed`` recognises it after probing dataset[i] against the array view -- and the .bin / labels are
    the bytes the DataLoader path writes (forced here by asking for workers).  A contiguous Subset is a slice, a
    ConcatDataset a list of segments; a dataset whose __getitem__ does anything else fails the probe and keeps the
    loader path."""
    import hubconf
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from workloads import Stl10Shaped
    comp, transform = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic", gpu_preprocess=True)
    ds = Stl10Shaped(700, transform)
    assert comp._array_backed(ds) is not None
    fa, fl = tmp_path / "a.bin", tmp_path / "l.bin"
    ya, yl = tmp_path / "a.npy", tmp_path / "l.npy"
    comp.compress_dataset(ds, str(fa), label_file=str(ya), is_info=False)                      # the README call
    comp.compress_dataset(ds, str(fl), label_file=str(yl), is_info=False,
                          kwargs_dataloader=dict(batch_size=128, num_workers=2))                 # DataLoader workers
    assert hashlib.sha256(fa.read_bytes()).hexdigest() == hashlib.sha256(fl.read_bytes()).hexdigest()
    assert np.array_equal(np.load(ya), np.load(yl)) and np.array_equal(np.load(ya), ds.labels.astype(np.uint16))
    z, y = comp.decompress_dataset(str(fa), label_file=str(ya), is_info=False)
    assert np.array_equal(y, ds.labels)
    x5 = [ds[i][0] for i in range(5)]
    assert np.array_equal(z[:5], comp(x5).cpu().numpy())

    sub = torch.utils.data.Subset(ds, range(100, 333))
    cat = torch.utils.data.ConcatDataset([sub, ds])
    for d in (sub, cat):
        assert comp._array_backed(d) is not None
        comp.compress_dataset(d, str(fa), label_file=str(ya), is_info=False)
        comp.compress_dataset(d, str(fl), label_file=str(yl), is_info=False,
                              kwargs_dataloader=dict(batch_size=100, num_workers=2))
        assert fa.read_bytes() == fl.read_bytes() and np.array_equal(np.load(ya), np.load(yl))

    class Mirrored(Stl10Shaped):                      # same arrays, another __getitem__: must NOT take the array view
        def __getitem__(self, i):
            x, t = super().__getitem__(i)
            return x.flip(1), t

    class OtherTargetName(Stl10Shaped):               # samples carry a target the array view has no array for
        def __init__(self, n, tf):
            super().__init__(n, tf)
            self.y, self.labels = self.labels, None

        def __getitem__(self, i):
            from PIL import Image
            return self.transform(Image.fromarray(np.transpose(self.data[i], (1, 2, 0)))), int(self.y[i])

    odd = OtherTargetName(200, transform)
    assert comp._array_backed(odd) is None
    comp.compress_dataset(odd, str(fa), label_file=str(ya), is_info=False)
    assert np.array_equal(np.load(ya), odd.y.astype(np.uint16))

    class FewRelabelled(Stl10Shaped):                 # first / middle / last samples agree, 5 % of the others do not
        def __getitem__(self, i):
            x, t = super().__getitem__(i)
            return x, (t + 1) % 10 if (i % 20 == 7) else t

    noisy = FewRelabelled(2000, transform)
    assert comp._array_backed(noisy) is None
    comp.compress_dataset(noisy, str(fa), label_file=str(ya), is_info=False)
    assert np.array_equal(np.load(ya), np.array([noisy[i][1] for i in range(2000)], dtype=np.uint16))

    # loader arguments that change what is loaded are the DataLoader's business: a sampler reverses the order here
    rev = list(range(len(ds) - 1, -1, -1))
    comp.compress_dataset(ds, str(fa), label_file=str(ya), is_info=False,
                          kwargs_dataloader=dict(batch_size=128, num_workers=0, sampler=rev))
    assert np.array_equal(np.load(ya), ds.labels[::-1].astype(np.uint16))

    aug = Mirrored(300, transform)
    assert comp._array_backed(aug) is None and comp._array_backed(torch.utils.data.Subset(ds, [3, 1, 2])) is None
    comp.compress_dataset(aug, str(fa), is_info=False)
    plain = Stl10Shaped(300, transform)
    comp.compress_dataset(plain, str(fl), is_info=False)
    assert fa.read_bytes() != fl.read_bytes()
    za = comp.decompress_dataset(str(fa), is_info=False)
    assert np.array_equal(za[:4], comp([aug[i][0] for i in range(4)]).cpu().numpy())
