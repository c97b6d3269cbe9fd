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
            img = pil_resize_crop_rgb(img, self.n_px)
            t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
        return (t - self.mean) / self.std


def pil_resize_crop_rgb(img, n_px=RES):
    """CLIP's PIL chain in CLIP's order -- Resize(n_px, bicubic) -> CenterCrop(n_px) -> convert("RGB") (clip.py
    ``_transform``) -- for a PIL image of any mode.  The order matters for everything that is not RGB or L: Pillow
    resizes RGBA premultiplied, palette images with nearest-neighbour, CMYK in four channels."""
    from PIL import Image
    w, h = img.size
    nw, nh = resized_size(w, h, n_px)
    img = img.resize((nw, nh), Image.BICUBIC)
    left, top = crop_origin(nw, nh, n_px)
    return img.crop((left, top, left + n_px, top + n_px)).convert("RGB")


def pillow_bicubic_taps(in_size, out_size, first, count):
    """Pillow ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for output positions
    ``first .. first+count-1`` of a resize ``in_size -> out_size`` (``lla_pillow_bicubic_taps``, the same
    double arithmetic as Pillow's Resample.c):
    (bounds int32 [count,2] = (first tap, number of taps), coef int32 [count,ksize])."""
    L = _lib.lib()
    ksize = int(L.lla_pillow_bicubic_ksize(in_size, out_size))
    if ksize <= 0:
        raise ValueError(f"bad resize {in_size} -> {out_size}")
    bounds = np.zeros((count, 2), np.int32)
    coef = np.zeros((count, ksize), np.int32)
    rc = L.lla_pillow_bicubic_taps(in_size, out_size, first, count, ksize,
                                   bounds.ctypes.data_as(ctypes.c_void_p), coef.ctypes.data_as(ctypes.c_void_p))
    _lib.check(rc, "lla_pillow_bicubic_taps")
    return bounds, coef


class RawRGB:
    """The ``transform`` of a compressor built with ``gpu_preprocess=True``: PIL image / HWC uint8 array ->
    uint8 tensor [H,W,3], untouched pixels.  Resize, crop, ToTensor and Normalize then happen on the GPU
    inside ``compress_dataset`` / ``compressor(X)`` (``ClipPreprocessGPU``: same bytes as the PIL chain), so
    the unchanged reference call ``STL10(transform=transform)`` -> ``compress_dataset(dataset, ...)``
    (hub/compressor.py:150-207, README) no longer spends its time in PIL resizes."""

    def __init__(self, n_px=RES):
        self.n_px = n_px

    def __call__(self, img):
        if isinstance(img, torch.Tensor):
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[-1] != 3:
                raise ValueError("expected a PIL image, an HWC uint8 array or a uint8 [H,W,3] tensor")
            return img
        if not isinstance(img, np.ndarray):
            if img.mode not in ("RGB", "L"):
                # RGBA / P / CMYK / ... (ImageNet holds a few CMYK JPEGs): the reference converts AFTER resize and crop,
                # and for these modes the order changes the pixels.  They take the PIL chain here; the GPU chain then
                # sees an n_px x n_px RGB image, on which its resize and crop are the identity.
                img = pil_resize_crop_rgb(img, getattr(self, "n_px", RES))
            img = np.array(img if img.mode == "RGB" else img.convert("RGB"), dtype=np.uint8)   # (a writable copy)
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise ValueError("expected an RGB uint8 image")
        return torch.from_numpy(np.ascontiguousarray(img))

    def __repr__(self):
        return "RawRGB()"


def _batch_bytes(n):
    """uint8 [n] for a collated batch; inside a DataLoader worker it is allocated in shared memory straight away (as
    torch's default_collate does), so that handing the batch to the main process does not copy it once more."""
    import torch.utils.data as tud
    if tud.get_worker_info() is not None:
        return torch.empty(0, dtype=torch.uint8).set_(torch.UntypedStorage._new_shared(int(n)))
    return torch.empty(int(n), dtype=torch.uint8)


class RaggedImages:
    """A batch of RGB uint8 images of different sizes: ``blob`` holds the pixels back to back (HWC per image,
    4 readable bytes of slack at the end), ``shapes`` int64 [B,2] = (H, W), ``offsets`` int64 [B]."""

    def __init__(self, blob, shapes, offsets):
        self.blob, self.shapes, self.offsets = blob, shapes, offsets

    @classmethod
    def from_list(cls, images):
        shapes = np.array([[int(t.shape[0]), int(t.shape[1])] for t in images], dtype=np.int64).reshape(-1, 2)
        sizes = shapes[:, 0] * shapes[:, 1] * 3
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(images) else np.zeros(0, np.int64)
        blob = _batch_bytes(int(sizes.sum()) + 4)
        blob[-4:] = 0
        for t, o, n in zip(images, offsets, sizes):
            blob[int(o):int(o + n)] = t.reshape(-1)
        return cls(blob, shapes, offsets)

    def __len__(self):
        return int(self.shapes.shape[0])

    @property
    def is_cuda(self):
        return self.blob.is_cuda

    def pin_memory(self):
        return RaggedImages(self.blob.pin_memory(), self.shapes, self.offsets)

    def to(self, device, non_blocking=False):
        return RaggedImages(self.blob.to(device, non_blocking=non_blocking), self.shapes, self.offsets)

    def image(self, i):
        """-> uint8 [H,W,3] view of image i."""
        h, w = (int(v) for v in self.shapes[i])
        o = int(self.offsets[i])
        return self.blob[o:o + h * w * 3].view(h, w, 3)


def ragged_collate(batch):
    """``collate_fn`` for datasets whose transform is :class:`RawRGB`: samples ``(uint8 [H,W,3], y, ...)`` ->
    ``(uint8 [B,H,W,3] or RaggedImages, collated y, ...)``; anything else goes to torch's default collate."""
    from torch.utils.data import default_collate
    first = batch[0][0] if isinstance(batch[0], (tuple, list)) else batch[0]
    if not (isinstance(first, torch.Tensor) and first.dtype == torch.uint8 and first.dim() == 3
            and first.shape[-1] == 3):
        return default_collate(batch)
    if isinstance(batch[0], (tuple, list)):
        imgs = [s[0] for s in batch]
        rest = [default_collate([s[k] for s in batch]) for k in range(1, len(batch[0]))]
    else:
        imgs, rest = list(batch), None
    if all(t.shape == imgs[0].shape for t in imgs):
        x = torch.stack(imgs, out=_batch_bytes(len(imgs) * imgs[0].numel()).view(len(imgs), *imgs[0].shape))
    else:
        x = RaggedImages.from_list(imgs)
    return x if rest is None else [x] + rest


_BANDS = (28, 14, 7, 4, 2, 1)       # band heights (output rows per workgroup) the ragged kernel is offered
_LDS_PREFERRED = 64 * 1024          # at least two workgroups per CU
_LDS_LIMIT = 160 * 1024             # one workgroup's LDS on gfx950


class ClipPreprocessGPU:
    """uint8 images on the GPU -> fp16 NHWC [B,224,224,3], same bytes as the PIL chain (``ClipPreprocess``
    followed by ``.half()``).  Accepts a uniform batch uint8 [B,H,W,3] (``lla_preprocess_clip``), a
    :class:`RaggedImages` batch or a list of uint8 [H,W,3] tensors (``lla_preprocess_clip_ragged``: every
    image its own size, tap tables cached per distinct size)."""

    def __init__(self):
        self._tables = {}
        self._geoms = {}
        self._mean = (ctypes.c_float * 3)(*CLIP_MEAN)
        self._std = (ctypes.c_float * 3)(*CLIP_STD)

    def __reduce__(self):   # (ctypes arrays and per-device tap tables do not pickle: a copy starts with empty caches)
        return (ClipPreprocessGPU, ())

    @staticmethod
    def _host_tables(H, W):
        nw, nh = resized_size(W, H)
        left, top = crop_origin(nw, nh)
        hb, hk = pillow_bicubic_taps(W, nw, left, RES)
        vb, vk = pillow_bicubic_taps(H, nh, top, RES)
        return hb, hk, vb, vk

    def _get(self, H, W, dev):
        key = (H, W, str(dev))
        if key not in self._tables:
            hb, hk, vb, vk = self._host_tables(H, W)
            row0 = int(vb[:, 0].min())
            nrows = int((vb[:, 0] + vb[:, 1]).max()) - row0
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            self._tables[key] = dict(hb=t(hb), hk=t(hk), vb=t(vb), vk=t(vk), hks=hk.shape[1],
                                     vks=vk.shape[1], row0=row0, nrows=nrows)
        return self._tables[key]

    def _geom(self, H, W, dev):
        """Ragged-kernel view of a size: device tap tables (bounds ‖ coef per axis) + LDS need per band height."""
        key = (H, W, str(dev))
        g = self._geoms.get(key)
        if g is None:
            hb, hk, vb, vk = self._host_tables(H, W)
            ht = np.ascontiguousarray(np.concatenate([hb.reshape(-1), hk.reshape(-1)]).astype(np.int32))
            vt = np.ascontiguousarray(np.concatenate([vb.reshape(-1), vk.reshape(-1)]).astype(np.int32))
            L, P = _lib.lib(), lambda a: a.ctypes.data_as(ctypes.c_void_p)
            need = {th: int(L.lla_preprocess_ragged_lds_bytes(P(ht), hk.shape[1], P(vt), vk.shape[1], th))
                    for th in _BANDS}
            htd, vtd = torch.from_numpy(ht).to(dev), torch.from_numpy(vt).to(dev)
            g = self._geoms[key] = dict(ht=htd, vt=vtd, hks=int(hk.shape[1]), vks=int(vk.shape[1]), need=need,
                                        hp=htd.data_ptr(), vp=vtd.data_ptr())
        return g

    _DESC = np.dtype([("pixels", "<u8"), ("h_table", "<u8"), ("v_table", "<u8"), ("H", "<i4"), ("W", "<i4"),
                      ("h_ksize", "<i4"), ("v_ksize", "<i4")])     # struct lla_image_desc (40 bytes)

    def _ragged(self, rag, out):
        dev = rag.blob.device
        _lib.require_cuda(rag.blob, "images")
        B = len(rag)
        if out is None:
            out = torch.empty((B, RES, RES, 3), dtype=torch.float16, device=dev)
        if B == 0:
            return out
        base = rag.blob.data_ptr()
        desc = np.zeros(B, dtype=self._DESC)
        geoms, huge = {}, []
        for i in range(B):
            h, w = int(rag.shapes[i, 0]), int(rag.shapes[i, 1])
            g = geoms.get((h, w))
            if g is None:
                g = geoms[(h, w)] = self._geom(h, w, dev)
            if g["need"][1] > _LDS_LIMIT:
                huge.append(i)
            desc[i] = (base + int(rag.offsets[i]), g["hp"], g["vp"], h, w, g["hks"], g["vks"])
        if huge:      # photos too large for a one-row band in LDS: one at a time through the two-pass path
            keep = np.ones(B, bool)
            keep[huge] = False
            for i in huge:
                self(rag.image(i).unsqueeze(0), out=out[i:i + 1])
        else:
            keep = None
        sel = desc if keep is None else desc[keep]
        if len(sel):
            fits = [g for g in geoms.values() if g["need"][1] <= _LDS_LIMIT]
            need = {th: max(g["need"][th] for g in fits) for th in _BANDS}
            th = next((t for t in _BANDS if need[t] <= _LDS_PREFERRED), None)
            if th is None:
                th = next(t for t in _BANDS if need[t] <= _LDS_LIMIT)
            dst = out if keep is None else torch.empty((len(sel), RES, RES, 3), dtype=torch.float16, device=dev)
            d = torch.from_numpy(sel.view(np.uint8).reshape(-1).copy()).to(dev)
            rc = _lib.lib().lla_preprocess_clip_ragged(_lib.ptr(d), len(sel), th, need[th], self._mean, self._std,
                                                       _lib.ptr(dst), _lib.stream_ptr(dev))
            _lib.check(rc, "lla_preprocess_clip_ragged")
            d.record_stream(torch.cuda.current_stream(dev))
            if keep is not None:
                out[torch.from_numpy(np.nonzero(keep)[0]).to(dev)] = dst
        return out

    def __call__(self, images, out=None):
        if isinstance(images, (list, tuple)):
            images = RaggedImages.from_list(list(images))
            images = images.to(torch.device("cuda", torch.cuda.current_device()))
        if isinstance(images, RaggedImages):
            return self._ragged(images, out)
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
