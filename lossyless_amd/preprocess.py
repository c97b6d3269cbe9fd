"""CLIP preprocessing: the reference's PIL transform and its GPU twin.

``ClipPreprocess``     -- per-image, PIL: what ``compressor.preprocess`` is in the reference
                          (clip._transform: Resize(224, BICUBIC), CenterCrop(224), ToTensor,
                          Normalize; hub/compressor.py:39,162-165; utils/data/images.py:383-411).
``ClipPreprocessGPU``  -- batched, ``lla_preprocess_clip``: uint8 [B,H,W,3] on the GPU ->
                          fp16 NHWC [B,224,224,3], bit-identical to ``ClipPreprocess`` followed by
                          ``.half()`` (Pillow's 8-bit resampler is integer arithmetic; its tap
                          tables are rebuilt here in float64 the way Pillow's Resample.c does).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

RES = 224
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # lossyless/helpers.py:252
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)   # lossyless/helpers.py:260
_PRECISION_BITS = 22


def resized_size(w, h, size=RES):
    """torchvision ``Resize(int)``: the smaller edge becomes ``size``, the other
    ``int(size * long / short)``."""
    if w <= h:
        return size, int(size * h / w)
    return int(size * w / h), size


def crop_origin(w, h, size=RES):
    """torchvision ``CenterCrop``: ``int(round((dim - size) / 2.0))`` (Python rounding)."""
    return int(round((w - size) / 2.0)), int(round((h - size) / 2.0))


class ClipPreprocess:
    """PIL image / HWC uint8 array -> CLIP-normalised float32 tensor [3,224,224]."""

    def __init__(self, n_px=RES):
        self.n_px = n_px
        self.mean = torch.tensor(CLIP_MEAN).view(3, 1, 1)
        self.std = torch.tensor(CLIP_STD).view(3, 1, 1)

    def __call__(self, img):
        from PIL import Image
        if isinstance(img, torch.Tensor):  # already a [3,H,W] tensor in [0,1]
            t = img.float()
        else:
            if isinstance(img, np.ndarray):
                img = Image.fromarray(img)
            img = img.convert("RGB")
            w, h = img.size
            nw, nh = resized_size(w, h, self.n_px)
            img = img.resize((nw, nh), Image.BICUBIC)
            left, top = crop_origin(nw, nh, self.n_px)
            img = img.crop((left, top, left + self.n_px, top + self.n_px))
            t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
        return (t - self.mean) / self.std


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_bicubic_taps(in_size, out_size, first, count):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for output positions
    ``first .. first+count-1`` of a resize ``in_size -> out_size``:
    (bounds int32 [count,2] = (first tap, number of taps), coef int32 [count,ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((count, 2), np.int32)
    coef = np.zeros((count, ksize), np.int32)
    for i in range(count):
        center = (first + i + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([_bicubic((x + xmin - center + 0.5) / filterscale) for x in range(xmax)],
                     dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << _PRECISION_BITS)
            coef[i, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[i] = (xmin, xmax)
    return bounds, coef


class ClipPreprocessGPU:
    """uint8 [B,H,W,3] (GPU) -> fp16 NHWC [B,224,224,3], same bytes as the PIL chain."""

    def __init__(self):
        self._tables = {}
        self._mean = (ctypes.c_float * 3)(*CLIP_MEAN)
        self._std = (ctypes.c_float * 3)(*CLIP_STD)

    def _get(self, H, W, dev):
        key = (H, W, str(dev))
        if key not in self._tables:
            nw, nh = resized_size(W, H)
            left, top = crop_origin(nw, nh)
            hb, hk = pillow_bicubic_taps(W, nw, left, RES)
            vb, vk = pillow_bicubic_taps(H, nh, top, RES)
            row0 = int(vb[:, 0].min())
            nrows = int((vb[:, 0] + vb[:, 1]).max()) - row0
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            self._tables[key] = dict(hb=t(hb), hk=t(hk), vb=t(vb), vk=t(vk), hks=hk.shape[1],
                                     vks=vk.shape[1], row0=row0, nrows=nrows)
        return self._tables[key]

    def __call__(self, images, out=None):
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError("expected uint8 [B,H,W,3]")
        images = images.contiguous()
        _lib.require_cuda(images, "images")
        B, H, W, _ = images.shape
        L = _lib.lib()
        t = self._get(H, W, images.device)
        wsb = int(L.lla_preprocess_workspace_bytes(B, t["nrows"]))
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=images.device)
        if out is None:
            out = torch.empty((B, RES, RES, 3), dtype=torch.float16, device=images.device)
        rc = L.lla_preprocess_clip(_lib.ptr(images), B, H, W, t["row0"], t["nrows"], _lib.ptr(t["hb"]),
                                   _lib.ptr(t["hk"]), t["hks"], _lib.ptr(t["vb"]), _lib.ptr(t["vk"]),
                                   t["vks"], self._mean, self._std, _lib.ptr(ws), wsb, _lib.ptr(out),
                                   _lib.stream_ptr(images.device))
        _lib.check(rc, "lla_preprocess_clip")
        return out
