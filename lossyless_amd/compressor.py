"""``ClipCompressor`` -- drop-in for ``hub/compressor.py`` on MI355X.

Same public surface as the reference class (hub/compressor.py:16-254): ``compressor(X)``,
``compress``, ``decompress``, ``get_rate``, ``compress_dataset``, ``decompress_dataset``,
``process_z_in/out``, ``.to(device)``, attributes ``preprocess / clip / z_dim / scaling /
biasing / entropy_bottleneck / device``, the same state-dict keys, the same ``.bin``
container and the same printed lines.  What differs is where the work runs: the CLIP
tower is one ``lla_vit_b32_forward`` call and the whole batch is quantised and rANS-coded
by ``lla_quantise_encode`` -- the reference's per-image Python loop is
gone.  Keyword-only extras: NHWC / device-tensor inputs and image-parallel sharding.
"""
import ctypes
import struct
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .clip_vit import VisionTransformer, resolve_clip_weights
from .preprocess import ClipPreprocess, ClipPreprocessGPU, RaggedImages, RawRGB, ragged_collate
from .entropy import EntropyBottleneck, update_registered_buffers
from . import distributed as lla_dist

try:  # progress bar as in the reference (hub/compressor.py:186); optional
    import tqdm as _tqdm
    _progress = _tqdm.tqdm
except Exception:  # pragma: no cover
    _progress = lambda it, **k: it


# The reference's default loader arguments (hub/compressor.py:154).  With ``gpu_preprocess=True`` and THESE defaults
# (the caller passed nothing) datasets of up to 12 288 images are loaded in the main process instead: the per-image
# work left on the host is a pixel copy (~12k img/s on one thread), and 16 workers take 0.7-0.9 s to start; up to
# 262 144 images 8 workers are started (enough to feed one tower; half the start-up).
import os as _os
_TOWER_BATCH = int(_os.environ.get("LLA_TOWER_BATCH", "8704"))    # images per tower pass RecordStream gathers (= the library's default slice, csrc/switches_product.cpp default_chunk)
# First tower passes of a call whose images start in HOST memory (images each; multiples of 128 = whole 256-row GEMM tiles):
# the tower starts after the first 1024 images have crossed the bus instead of after a whole pass of 8704 (35 ms of
# staging at STL10's image size), and every later, larger pass is staged under the one before it.  LLA_TOWER_RAMP=0: none.
_LIB_SLICE = 8704       # csrc/switches_product.cpp default_chunk(): images per library slice (one in-place pass must fit in one)
_TOWER_RAMP = tuple(int(v) for v in _os.environ.get("LLA_TOWER_RAMP", "1024,2176,4352").split(",") if int(v) > 0)
_DEFAULT_LOADER = dict(batch_size=128, num_workers=16)
_INLINE_LOADER_MAX = 12288
_FEW_WORKERS_MAX = 262144

_MADV_DONTFORK, _MADV_DOFORK = 10, 11    # <linux/mman.h>
_libc = None


def _fork_advice(t, advice):
    """madvise() a page-aligned pinned host tensor.  Where ROCr backs pinned memory with userptr pages (its default;
    ``lossyless_amd/__init__.py`` asks for GTT buffers instead when it is imported before the HIP runtime starts),
    every fork() write-protects them, the kernel driver evicts the process's GPU queues and re-pins the lot: each
    DataLoader worker started costs a GPU stall proportional to the pinned bytes (measured on the MI355X host:
    a 16-worker call 0.7 s -> 11.6 s with one pinned GiB, tools/fork_probe3.py, tools/h2d_probe.py).  The staging
    buffers of ``_prefetch`` are private to this module -- a worker never reads them -- so they are kept out of
    the children altogether."""
    global _libc
    n = t.numel() * t.element_size()
    if t.data_ptr() % 4096 or n < 65536:
        return
    try:
        if _libc is None:
            _libc = ctypes.CDLL(None, use_errno=True)
            _libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _libc.madvise(ctypes.c_void_p(t.data_ptr()), (n + 4095) // 4096 * 4096, advice)
    except (OSError, AttributeError):  # pragma: no cover
        pass


class _PinnedStaging:
    """Two growing pinned host buffers (one being filled while the other's copy is in flight), owned by one
    compressor for its lifetime: pinning a buffer costs ~0.4 s per GB, so they are not re-made per call, and
    they are MADV_DONTFORK while this object holds them (``_fork_advice``)."""

    def __init__(self):
        self.buf = [None, None]

    def get(self, k, numel, dtype):
        buf = self.buf[k]
        if buf is None or buf.dtype != dtype or buf.numel() < numel:
            if buf is not None:
                _fork_advice(buf, _MADV_DOFORK)   # (goes back to torch's host allocator: an ordinary block again)
            self.buf[k] = buf = None
            buf = self.buf[k] = torch.empty(max(numel, 1) * (5 if numel > (1 << 20) else 4) // 4, dtype=dtype).pin_memory()
            _fork_advice(buf, _MADV_DONTFORK)
        return buf[:numel]

    def __reduce__(self):   # (a pickled / deep-copied compressor starts with none)
        return (_PinnedStaging, ())

    def trim(self, keep_bytes=256 << 20):
        """End of a compress_dataset call: buffers above ``keep_bytes`` go back to the allocator (re-pinning costs
        ~0.4 s per GB on the next large call; holding 2 x 1.6 GB of pinned, MADV_DONTFORK host memory for the
        lifetime of the compressor after ONE call on a large host tensor costs the process more)."""
        for k, b in enumerate(self.buf):
            if b is not None and b.numel() * b.element_size() > keep_bytes:
                _fork_advice(b, _MADV_DOFORK)
                self.buf[k] = None

    def __del__(self):
        for b in self.buf:
            if b is not None:
                _fork_advice(b, _MADV_DOFORK)


class ClipCompressor(nn.Module):
    """CLIP ViT-B/32 compressor (see the reference docstring, hub/compressor.py:17-30).

    Parameters
    ----------
    pretrained_state_dict : dict or str or Path
        State dict of the rate estimator (``scaling``, ``biasing``, ``entropy_bottleneck.*``)
        or a path to it.
    is_jit : bool
        Accepted for signature compatibility; there is no TorchScript on this path.
    device : str
        ``"cuda"`` (an MI355X).  ``"cpu"`` builds the module (tables, weights) but every
        compute entry point raises: there is no CPU fallback.
    clip_weights : None, "synthetic", path or dict, keyword-only
        CLIP ViT-B/32 visual weights (``clip.load`` is unavailable offline): a path to the
        OpenAI checkpoint / a state-dict, default ``$LOSSYLESS_CLIP_WEIGHTS``; ValueError when
        neither is given.  ``"synthetic"`` (seed-1 random weights) must be asked for by name.
    vit_chunk : int, keyword-only
        Images per slice inside the tower (0 = library default).
    gpu_preprocess : bool, keyword-only
        False (default): ``self.preprocess`` is the reference's PIL chain (resize, centre crop, ToTensor,
        Normalize per image on the host).  True: ``self.preprocess`` only hands the raw RGB pixels over
        (:class:`RawRGB`) and the same chain runs on the GPU, bit-identically, inside ``compress_dataset`` /
        ``compressor(X)``; images of different sizes are batched by ``ragged_collate``.
    """

    def __init__(self, pretrained_state_dict, is_jit=False,
                 device="cuda" if torch.cuda.is_available() else "cpu", *,
                 clip_weights=None, vit_chunk=0, gpu_preprocess=False):
        super().__init__()
        vit_sd, self.clip_weights_desc = resolve_clip_weights(clip_weights)
        self.clip = VisionTransformer(vit_sd, chunk=vit_chunk)
        self.gpu_preprocess = bool(gpu_preprocess)
        self._staging = _PinnedStaging()
        self.preprocess = RawRGB() if self.gpu_preprocess else ClipPreprocess()
        self.preprocess_gpu = ClipPreprocessGPU()   # batched twin: uint8 images -> fp16 NHWC

        self.z_dim = 512
        self.side_z_dim = 512 // 5

        self.scaling = torch.nn.Parameter(torch.ones(self.z_dim))
        self.biasing = torch.nn.Parameter(torch.zeros(self.z_dim))

        self.entropy_bottleneck = EntropyBottleneck(self.z_dim, init_scale=10, filters=[3, 3, 3, 3])

        if not isinstance(pretrained_state_dict, dict):
            # the reference reads an undefined name here (hub/compressor.py:53-54)
            pretrained_state_dict = torch.load(pretrained_state_dict, map_location="cpu",
                                               weights_only=True)

        update_registered_buffers(
            self.entropy_bottleneck, "entropy_bottleneck",
            ["_quantized_cdf", "_offset", "_cdf_length"], pretrained_state_dict)
        self.load_state_dict(pretrained_state_dict, strict=False)
        self.entropy_bottleneck.update()  # no-op when the state dict carried frozen tables

        self.device = device
        self.to(self.device)
        self.eval()

    def to(self, device):
        self.device = device
        return super().to(device)

    # ------------------------------------------------------------------ helpers
    def _tables(self):
        return self.entropy_bottleneck.device_tables(self.scaling, self.biasing)

    def _check_gpu(self):
        if str(self.device) == "cpu" or not torch.cuda.is_available():
            raise RuntimeError("lossyless_amd computes on MI355X only: move the compressor to "
                               "'cuda' (there is no CPU fallback)")

    def _embed(self, X):
        self._check_gpu()
        if isinstance(X, (list, tuple)):        # raw RGB images of different sizes
            X = RaggedImages.from_list(list(X))
        if not X.is_cuda:
            X = X.to(self.device)
        if isinstance(X, RaggedImages) or X.dtype == torch.uint8:
            X = self.preprocess_gpu(X)          # raw RGB: resize / crop / normalise on the GPU
        return self.clip(X)

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def forward(self, X, is_compress=False):
        """Featurise a batch with or without compression (hub/compressor.py:73-103).

        X : [B,3,224,224] (or [B,224,224,3]) CLIP-normalised images.
        Returns a list of ``bytes`` if ``is_compress`` else z_hat [B,512] fp32."""
        z = self._embed(X)
        tables = self._tables()
        if is_compress:
            payload, offsets, _ = self.entropy_bottleneck.encode_device(z, tables)
            off = offsets.cpu().numpy()
            blob = payload[: int(off[-1])].cpu().numpy().tobytes()
            return [blob[int(off[i]):int(off[i + 1])] for i in range(z.shape[0])]
        B, C = z.shape
        out = torch.empty((B, C), dtype=torch.float32, device=z.device)
        rc = _lib.lib().lla_represent(_lib.ptr(z), _lib.LLA_Z_F16, B, C, _lib.ptr(tables["bias"]),
                                      _lib.ptr(tables["exp_scale"]), _lib.ptr(tables["median"]),
                                      _lib.ptr(out), _lib.stream_ptr(z.device))
        _lib.check(rc, "lla_represent")
        return out

    def process_z_in(self, z):
        """(z.float() + biasing) * exp(scaling), as [B,512,1,1] (hub/compressor.py:105-109)."""
        t = self._tables()
        z_in = (z.float() + t["bias"].to(z.device)) * t["exp_scale"].to(z.device)
        return z_in.unsqueeze(-1).unsqueeze(-1)

    def process_z_out(self, z_hat):
        """z_hat / exp(scaling) - biasing (hub/compressor.py:111-115)."""
        t = self._tables()
        z_hat = z_hat.squeeze(-1).squeeze(-1)
        return (z_hat / t["exp_scale"].to(z_hat.device)) - t["bias"].to(z_hat.device)

    def compress(self, X):
        """Return compressed features (list of byte strings), hub/compressor.py:117-119."""
        return self(X, is_compress=True)

    @torch.no_grad()
    def decompress(self, byte_str):
        """Decompress byte strings -> z_hat [B,512] fp32 on ``self.device`` (hub/compressor.py:121-125).

        On a module moved to the CPU (what the reference's ``decompress_dataset(is_cpu=True)`` does before
        calling this, hub/compressor.py:227-229) the strings are decoded by the library's HOST coder
        (``lla_rans_decode_batch_host``); on the GPU by ``lla_rans_decode_batch``.  Same values bit for bit."""
        if str(self.device) == "cpu":
            body, off = self._records_of(byte_str)
            return torch.from_numpy(self._decode_records_host(body, off, len(byte_str)))
        self._check_gpu()
        return self._decode_strings(byte_str)

    def get_rate(self, X):
        """Mean coded size per image in bits (hub/compressor.py:127-135)."""
        byte_str = self.compress(X)
        n_bytes = sum([len(s) for s in byte_str]) / len(byte_str)
        return n_bytes * 8

    def make_pickable_(self):
        """No coder object is held (the C-ABI is stateless), so the module always pickles."""

    def undo_pickable_(self):
        """See ``make_pickable_``."""

    # ------------------------------------------------------------------ decode core
    def _decode_records(self, body, off_np, B):
        """body: uint8 numpy of be32-prefixed records; off_np: uint64 [B+1] -> fp32 [B,512] tensor."""
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        tables = self._tables()
        pad = (-len(body)) % 4 + 4
        payload = torch.from_numpy(np.concatenate([body, np.zeros(pad, np.uint8)])).to(dev)
        offsets = torch.from_numpy(off_np.astype(np.int64)).to(dev)
        sym, status = self.entropy_bottleneck.decode_device(payload, offsets, B, tables,
                                                            record_prefix=True)
        out = torch.empty((B, self.z_dim), dtype=torch.float32, device=dev)
        rc = _lib.lib().lla_dequantise(_lib.ptr(sym), B, self.z_dim, _lib.ptr(tables["bias"]),
                                       _lib.ptr(tables["exp_scale"]), _lib.ptr(tables["median"]),
                                       _lib.ptr(out), _lib.stream_ptr(dev))
        _lib.check(rc, "lla_dequantise")
        if B and int(status.max()) != 0:
            raise ValueError("malformed rANS stream in container")
        return out

    @staticmethod
    def _records_of(strings):
        """list[bytes] -> (container body uint8, record offsets uint64 [B+1])."""
        B = len(strings)
        parts, off = [], np.zeros(B + 1, dtype=np.uint64)
        for i, s in enumerate(strings):
            parts.append(struct.pack(">I", len(s)))
            parts.append(bytes(s))
            off[i + 1] = off[i] + 4 + len(s)
        return np.frombuffer(b"".join(parts), dtype=np.uint8), off

    def _decode_strings(self, strings):
        body, off = self._records_of(strings)
        return self._decode_records(body, off, len(strings))

    # ------------------------------------------------------------------ datasets
    @torch.no_grad()
    def compress_dataset(self, dataset, file, label_file=None,
                         kwargs_dataloader=_DEFAULT_LOADER, is_info=True, *,
                         distributed=False, entropy_group=16, coalesce=_TOWER_BATCH):
        """Compress a dataset and save it to ``file`` (hub/compressor.py:150-207).

        ``dataset`` is a map-style dataset yielding ``(x[3,224,224], y, ...)`` exactly as in
        the reference, or -- fast path -- a tensor of images ([N,3,224,224] or NHWC
        [N,224,224,3]; on the GPU it is sliced in place, no DataLoader).  With
        ``distributed=True`` under an initialised ``torch.distributed`` group, every rank
        encodes a contiguous shard and rank 0 writes a file byte-identical to the 1-GPU one.
        ``entropy_group``: how many thousand (1024) images' embeddings are entropy-coded together (see
        :class:`RecordStream`); ``coalesce``: batches smaller than this many images are gathered into tower
        batches of that size (default 8704 = 1700 row tiles of 256: the persistent GEMMs' rounds come out 99.6 % full on
        256 CUs and there are 4x fewer launches -- 99.5k vs 94.9k img/s for the tower alone against 1024-image
        batches; 0: the tower runs once per batch as given).  Any values give the same file.
        """
        if str(self.device) == "cpu":
            raise ValueError("Compression only implemented on GPU (as uses fp16).")
        self._check_gpu()

        start = time.time()
        rank, world = (lla_dist.rank_world() if distributed else (0, 1))
        n_total = len(dataset)
        lo, hi = lla_dist.shard_bounds(n_total, rank, world)

        # in-memory array datasets are read straight from their array unless the caller asked for loader workers
        # (only when the loader arguments cannot change WHAT is loaded: a sampler, shuffle, drop_last or a collate_fn
        # is the DataLoader's business, hub/compressor.py:155)
        plain_loader = kwargs_dataloader is _DEFAULT_LOADER or (
            not kwargs_dataloader.get("num_workers", 0) and
            set(kwargs_dataloader) <= {"batch_size", "num_workers", "pin_memory", "prefetch_factor", "persistent_workers"})
        arrays = self._array_backed(dataset) if plain_loader else None
        if arrays is None and kwargs_dataloader is _DEFAULT_LOADER and self.gpu_preprocess and not isinstance(dataset, torch.Tensor):
            # the caller passed no loader arguments: the per-image host work is a pixel copy (~13k img/s per process on the
            # MI355X host) and every worker costs 20-45 ms of fork() before the first batch -- none for small datasets,
            # 8 (what it takes to feed one tower) up to a few seconds' worth of images, the reference's 16 beyond
            if hi - lo <= _INLINE_LOADER_MAX:
                kwargs_dataloader = dict(_DEFAULT_LOADER, num_workers=0)
            elif hi - lo <= _FEW_WORKERS_MAX:
                kwargs_dataloader = dict(_DEFAULT_LOADER, num_workers=8)
        # images that start on the host: short first tower passes (see _TOWER_RAMP); device-resident / generated-on-device
        # data has nothing to wait for and goes in whole passes from the start
        on_host = not (hasattr(dataset, "device_batch") or (isinstance(dataset, torch.Tensor) and dataset.is_cuda))
        ramp = tuple(r for r in _TOWER_RAMP if r < int(coalesce)) if (coalesce and on_host) else ()
        # under nccl the ranks that only SEND keep their records on the GPU (distributed.gather_bytes_to_rank0)
        stream = self.record_stream(entropy_group, coalesce, ramp, on_device=world > 1 and lla_dist.sends_from_device(self.device))
        Y, n_local = [], 0
        if arrays is not None:
            kwargs_dataloader = dict(kwargs_dataloader, batch_size=max(int(kwargs_dataloader.get("batch_size", 128)),
                                                                       int(coalesce) or 1024))
        if coalesce and (isinstance(dataset, torch.Tensor) or hasattr(dataset, "device_batch")):
            # data that is sliced / generated on demand comes in tower-pass-sized pieces straight away: nothing to gather
            kwargs_dataloader = dict(kwargs_dataloader,
                                     batch_size=max(int(kwargs_dataloader.get("batch_size", 128)), int(coalesce)))
        batches = (self._array_batches(arrays, lo, hi, int(kwargs_dataloader["batch_size"]), label_file is not None, ramp)
                   if arrays is not None else
                   self._batches(dataset, lo, hi, kwargs_dataloader, label_file is not None, ramp))
        planar = arrays is not None and arrays[0][1] == "chw"
        # Host side of the loop (collation in the main process when num_workers=0, fp32 -> fp16 staging): torch's
        # intra-op pool defaults to one thread per hardware thread, and on a 256-thread GPU host `torch.stack` of a
        # 77 MB batch then takes seconds (measured: 8 img/s with 256 threads, 8.8k img/s with 4).
        host_threads = torch.get_num_threads()
        if host_threads > _HOST_THREADS:
            torch.set_num_threads(_HOST_THREADS)
        try:
            for x, y in self._prefetch(batches):
                if planar:     # torchvision's STL10 / SVHN keep [N,3,H,W]: interleave on the device (a copy kernel)
                    x = x.permute(0, 2, 3, 1).contiguous()
                # (the batches of this loop are made here -- staged copies, generated images -- or are slices of the
                # caller's device tensor, which nobody writes during the call: the stream may read them in place)
                stream.push(x, donate=True)
                n_local += len(x)
                if y is not None:
                    Y += [y.cpu().numpy().astype(np.uint16)]
        finally:
            if host_threads > _HOST_THREADS:
                torch.set_num_threads(host_threads)

        body = stream.finish()
        self._staging.trim()
        labels = np.concatenate(Y) if Y else np.zeros(0, np.uint16)
        if world > 1:
            body, labels, n_all = lla_dist.gather_to_rank0(body, labels, n_local, self.device)
        else:
            n_all = n_local

        if rank == 0:
            with Path(file).open("wb") as f:
                f.write(struct.pack(">I", n_all))   # write_uints(f, (len(Z_bytes),))
                f.write(body.tobytes())             # N x { >I len, bytes }
            enc_time = (time.time() - start) / max(n_all, 1)
            rate = 8 * Path(file).stat().st_size / max(n_all, 1)
            if label_file is not None:
                np.save(label_file, labels, allow_pickle=False)  # no pickle for portability
            if is_info:
                print(f"Rate: {rate:.2f} bits/img | Encoding: {1/enc_time:.2f} img/sec ")
        if world > 1:
            lla_dist.barrier()

    @torch.no_grad()
    def encode_batch_records(self, x):
        """One pass of the hot path over one batch: images -> CLIP tower -> quantise -> rANS ->
        compaction into container records (be32 length + stream per image), returned as a
        host uint8 array.  One device->host sync per batch (the reference syncs per image)."""
        z = self._embed(x)
        payload, offsets, _ = self.entropy_bottleneck.encode_device(z, self._tables(),
                                                                    record_prefix=True)
        total = int(offsets[-1])
        return payload[:total].cpu().numpy()

    def _prefetch(self, batches):
        """Host batches -> device batches, one batch ahead: batch i+1 is staged in pinned memory and
        copied on a side stream while the tower runs on batch i (the reference does a synchronous
        ``x.to(device).half()`` per batch, hub/compressor.py:187).  Device batches pass through."""
        dev = torch.device(self.device)
        copy_stream = None
        slot_event = [None, None]           # last copy issued out of each staging buffer
        pending = None                      # (device tensor, labels, event)
        k = 0
        for x, y in batches:
            if isinstance(x, RaggedImages) and not x.is_cuda:
                # ragged uint8 batch: the blob is one 1-D tensor and takes the same staged copy; sizes ride along
                shapes, offsets, x = x.shapes, x.offsets, x.blob
                rewrap = lambda t, s=shapes, o=offsets: RaggedImages(t, s, o)
            else:
                rewrap = None
            if x.is_cuda:
                if pending is not None:
                    torch.cuda.current_stream(dev).wait_event(pending[2])
                    yield (pending[3](pending[0]) if pending[3] else pending[0]), pending[1]
                    pending = None
                yield x, y
                continue
            if copy_stream is None:
                copy_stream = torch.cuda.Stream(device=dev)
            # the tower takes fp16: halve the bytes before the bus, in the same pass that stages the batch in
            # pinned memory (one read of the fp32 batch instead of a .half() and a copy)
            want = torch.float16 if x.dtype == torch.float32 else x.dtype
            if not x.is_pinned() or x.dtype != want:
                if slot_event[k] is not None:
                    slot_event[k].synchronize()   # its previous copy must have left the buffer
                buf = self._staging.get(k, x.numel(), want).view(x.shape)
                buf.copy_(x)
                x = buf
            with torch.cuda.stream(copy_stream):
                xd = x.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            slot_event[k] = ev
            if pending is not None:
                torch.cuda.current_stream(dev).wait_event(pending[2])
                pending[0].record_stream(torch.cuda.current_stream(dev))
                yield (pending[3](pending[0]) if pending[3] else pending[0]), pending[1]
            pending = (xd, y, ev, rewrap)
            k ^= 1
        if pending is not None:
            torch.cuda.current_stream(dev).wait_event(pending[2])
            pending[0].record_stream(torch.cuda.current_stream(dev))
            yield (pending[3](pending[0]) if pending[3] else pending[0]), pending[1]

    def record_stream(self, group=16, coalesce=_TOWER_BATCH, ramp=(), on_device=False):
        """-> :class:`RecordStream` over this compressor (what ``compress_dataset`` loops with)."""
        return RecordStream(self, group, coalesce, ramp, on_device)

    def _array_backed(self, dataset):
        """In-memory image datasets (torchvision's STL10 / CIFAR / SVHN and their look-alikes keep every image in ONE
        uint8 array ``.data`` -- [N,3,H,W] or [N,H,W,3] -- and ``__getitem__`` is ``(transform(Image.fromarray(data[i])),
        target)``): with ``gpu_preprocess=True`` the transform is :class:`RawRGB`, i.e. a sample IS ``data[i]``, and
        the per-image Python round trip (PIL object, tensor, collate: 75 us per image on the MI355X host = 13k img/s
        per process, GIL-bound so that threads make it slower, and 45 ms of fork() per worker process) is pure
        overhead.  Returns segments ``[(data, "chw" | "hwc", labels or None), ...]`` (one; several for a ``ConcatDataset``; a
        contiguous ``Subset`` is a slice) when ``dataset`` is such an object AND its own
        ``__getitem__`` agrees with the array view on probe samples (first, middle, last and up to 256 pseudo-random ones: pixels and label) -- a
        dataset that does anything else in ``__getitem__`` fails the probe and goes through the DataLoader -- else
        None.  The file is the same bytes either way (tests/test_gpu_configs.py)."""
        if not self.gpu_preprocess or isinstance(dataset, torch.Tensor) or hasattr(dataset, "device_batch"):
            return None
        from torch.utils.data import ConcatDataset, Subset
        if isinstance(dataset, Subset):     # a contiguous ascending run of an array-backed dataset (a split, a shard)
            idx = dataset.indices
            n = len(idx)
            if n == 0 or not all(int(idx[k]) == int(idx[0]) + k for k in (0, n // 2, n - 1)) or \
                    (not isinstance(idx, range) and list(map(int, idx)) != list(range(int(idx[0]), int(idx[0]) + n))):
                return None
            inner = self._array_backed(dataset.dataset)
            if inner is None or len(inner) != 1:
                return None
            data, layout, labels = inner[0]
            a, b = int(idx[0]), int(idx[0]) + n
            return [(data[a:b], layout, None if labels is None else labels[a:b])]
        if isinstance(dataset, ConcatDataset):
            parts = [self._array_backed(d) for d in dataset.datasets]
            if any(p is None for p in parts):
                return None
            segs = [seg for p in parts for seg in p]
            if len({(seg[1], seg[0].shape[1:], seg[2] is None) for seg in segs}) != 1:
                return None
            return segs
        data = getattr(dataset, "data", None)
        if (not isinstance(data, np.ndarray) or data.dtype != np.uint8 or data.ndim != 4 or len(data) != len(dataset)
                or not isinstance(getattr(dataset, "transform", None), RawRGB)
                or getattr(dataset, "target_transform", None) is not None):
            return None
        layout = "hwc" if data.shape[3] == 3 else ("chw" if data.shape[1] == 3 else None)
        if layout is None or len(data) == 0:
            return None
        labels = getattr(dataset, "labels", None)
        if labels is None:
            labels = getattr(dataset, "targets", None)
        if labels is not None:
            labels = np.asarray(labels)
            if labels.shape != (len(data),):
                return None
        try:
            # first, middle, last and up to 256 pseudo-random samples (a subclass that remaps or relabels SOME indices --
            # noisy-label CIFAR, an index permutation with fixed points -- must not slip through three probes)
            n = len(data)
            probes = {0, n // 2, n - 1} | {int(i) for i in np.random.default_rng(n).integers(0, n, size=min(n, 256))}
            for i in sorted(probes):
                sample = dataset[i]
                x = sample[0] if isinstance(sample, (tuple, list)) else sample
                want = data[i].transpose(1, 2, 0) if layout == "chw" else data[i]
                if not (isinstance(x, torch.Tensor) and x.dtype == torch.uint8 and tuple(x.shape) == want.shape
                        and np.array_equal(x.numpy(), want)):
                    return None
                if labels is not None and (not isinstance(sample, (tuple, list)) or len(sample) < 2
                                           or int(sample[1]) != int(labels[i])):
                    return None
                if labels is None and isinstance(sample, (tuple, list)) and len(sample) > 1:
                    return None     # the samples carry a target this view has no array for: the DataLoader serves it
        except Exception:
            return None
        return [(data, layout, labels)]

    @staticmethod
    def _array_batches(arrays, lo, hi, bs, want_labels, ramp=()):
        """Batches over images lo .. hi-1 of the concatenated segments (a batch ends at a segment boundary); the first
        batches have the sizes in ``ramp`` (short first tower passes, see _TOWER_RAMP)."""
        base, sizes = 0, list(ramp)
        for data, _, labels in arrays:
            a, b = max(lo - base, 0), min(hi - base, len(data))
            i = a
            while i < b:
                j = min(i + (min(sizes.pop(0), bs) if sizes else bs), b)
                x = torch.from_numpy(data[i:j])    # a view: staged into pinned memory by _prefetch, no per-image work
                y = torch.from_numpy(np.ascontiguousarray(labels[i:j])) if (want_labels and labels is not None) else None
                yield x, y
                i = j
            base += len(data)

    def _batches(self, dataset, lo, hi, kwargs_dataloader, want_labels, ramp=()):
        """Yield (x, y-or-None) over dataset[lo:hi]."""
        bs = int(kwargs_dataloader.get("batch_size", 128))
        if isinstance(dataset, torch.Tensor):
            i, sizes = lo, list(ramp)
            while i < hi:
                j = min(i + (min(sizes.pop(0), bs) if sizes else bs), hi)
                yield dataset[i:j], None
                i = j
            return
        if hasattr(dataset, "device_batch"):   # lazily generated device batches (see SyntheticImages)
            for i in range(lo, hi, bs):
                yield dataset.device_batch(i, min(i + bs, hi), self.device), None
            return
        from torch.utils.data import DataLoader, Subset
        ds = dataset if (lo == 0 and hi == len(dataset)) else Subset(dataset, range(lo, hi))
        if self.gpu_preprocess and "collate_fn" not in kwargs_dataloader:
            # samples are raw uint8 [H,W,3] images (RawRGB), possibly of different sizes
            kwargs_dataloader = dict(kwargs_dataloader, collate_fn=ragged_collate)
        for x, *y in _progress(DataLoader(ds, **kwargs_dataloader)):
            yield x, (y[0] if (want_labels and y) else None)

    def _decode_records_host(self, body, off_np, B):
        """Host twin of ``_decode_records`` (``lla_rans_decode_batch_host`` + ``lla_dequantise_host``):
        uint8 numpy records + uint64 offsets -> float32 [B,512] ndarray, no GPU involved."""
        t = self._tables()
        cdf = t["cdf"].cpu().numpy()
        cdf_len = t["cdf_len"].cpu().numpy()
        offset = t["offset"].cpu().numpy()
        bias, es, med = (t[k].cpu().numpy() for k in ("bias", "exp_scale", "median"))
        body = np.ascontiguousarray(body, dtype=np.uint8)
        off = np.ascontiguousarray(off_np, dtype=np.uint64)
        sym = np.empty((B, self.z_dim), dtype=np.int32)
        status = np.zeros(max(B, 1), dtype=np.int32)
        out = np.empty((B, self.z_dim), dtype=np.float32)
        L, P = _lib.lib(), lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = L.lla_rans_decode_batch_host(P(body), P(off), 1, B, self.z_dim, P(cdf), t["W"],
                                          P(cdf_len), P(offset), P(sym), P(status))
        _lib.check(rc, "lla_rans_decode_batch_host")
        if B and int(status.max()) != 0:
            raise ValueError("malformed rANS stream in container")
        rc = L.lla_dequantise_host(P(sym), B, self.z_dim, P(bias), P(es), P(med), P(out))
        _lib.check(rc, "lla_dequantise_host")
        return out

    @torch.no_grad()
    def decompress_dataset(self, file, label_file=None, is_info=True, is_cpu=True, *,
                           batch_size=65536):
        """Decompress a dataset saved on file and return a numpy array
        (hub/compressor.py:209-254).

        ``is_cpu=True`` (the reference's default: it moves the module to the host and decodes one
        image per Python iteration, :227-229,236-238) decodes with the library's HOST coder
        (``lla_rans_decode_batch_host``, threaded over images) and needs no GPU; ``is_cpu=False``
        decodes ``batch_size`` images at a time on the MI355X (``lla_rans_decode_batch``).  Both give
        the same float32 [N,512] ndarray, bit for bit.  Records are located by
        ``lla_container_index``."""
        if not is_cpu:
            self._check_gpu()
        start = time.time()

        blob = np.fromfile(str(file), dtype=np.uint8)
        L = _lib.lib()
        n = ctypes.c_uint32(0)
        # (validates the count against the file size before anything is sized by it)
        rc = L.lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size, None, 0,
                                   ctypes.byref(n))
        _lib.check(rc, "lla_container_index")
        n_Z = int(n.value)
        off = np.zeros(n_Z + 1, dtype=np.uint64)
        rc = L.lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                   off.ctypes.data_as(ctypes.c_void_p), off.size, ctypes.byref(n))
        _lib.check(rc, "lla_container_index")
        body = blob[4:]

        Z_hat = np.empty((n_Z, self.z_dim), dtype=np.float32)
        for i in range(0, n_Z, batch_size):
            j = min(i + batch_size, n_Z)
            b0, b1 = int(off[i]), int(off[j])
            if is_cpu:
                Z_hat[i:j] = self._decode_records_host(body[b0:b1], off[i:j + 1] - off[i], j - i)
            else:
                out = self._decode_records(body[b0:b1], off[i:j + 1] - off[i], j - i)
                Z_hat[i:j] = out.cpu().numpy()

        dec_time = (time.time() - start) / max(len(Z_hat), 1)
        if is_info:
            print(f"Decoding: {1/dec_time:.2f} img/sec ")

        if label_file is not None:
            Y = np.load(label_file, allow_pickle=False).astype(np.int64)
            return Z_hat, Y
        return Z_hat


_HOST_THREADS = 4      # torch intra-op threads during the host side of compress_dataset (see there)


class RecordStream:
    """Streaming encoder of ``compress_dataset``: ``push(images)`` runs the tower and parks the
    embeddings in a device buffer; every ``group`` x 1024 images (and at ``finish()``) the parked rows are
    quantised, rANS-coded and compacted into container records in ONE launch sequence with ONE
    device->host sync, and the bytes are appended to the output.

    Why: the coder walks a 512-symbol dependency chain per image with one image per lane, so a
    1024-image batch keeps 16 of the chip's 1024 SIMDs busy for ~150 us whatever the batch size
    up to 65536 images.  Coding 16 tower batches at once costs the same ~150 us once instead of
    16 times, and the per-batch host sync goes with it.  Records are position-independent, so
    the bytes are identical for every ``group`` (tests/test_gpu_compressor.py).

    Pipeline (nothing drains between groups): tower passes are queued on the library's two tower lanes
    without a join per batch (``lla_vit_b32_forward_deferred``: one batch's last GEMM rounds overlap the
    next batch's first kernels); a group's join + entropy coding run on a side ("coder") stream while
    the next group's tower passes are already being queued into the other embedding buffer, and a
    group's bytes are fetched one group later, when its kernels have long finished.  Pushed inputs
    and the embedding buffer stay referenced until their group has been fetched: the lanes and the
    coder stream use them outside the current stream's order."""

    def __init__(self, compressor, group=16, coalesce=_TOWER_BATCH, ramp=(), on_device=False):
        self.c = compressor
        self.group = max(int(group), 1)
        self.on_device = bool(on_device)     # finish() returns a device tensor (see there)
        # sizes of the first tower passes (smaller than `coalesce`; empty: whole passes from the start).  With images
        # arriving from the host, the first pass should not wait for `coalesce` of them (ClipCompressor.compress_dataset)
        self._ramp = [int(r) for r in ramp if 0 < int(r) < int(coalesce)]
        # Small pushes (the reference's default DataLoader batch is 128 images, BASELINE configs[0] uses 32) are
        # copied into a staging batch of `coalesce` images and the tower runs once per staging batch: its GEMMs
        # need ~50k rows per launch to fill the chip (tower alone: 19k / 48k / 75k / 93k img/s at batch 32 /
        # 128 / 256 / 1024).  Images are independent, so the records are the same bytes in the same order.
        self.coalesce = max(int(coalesce), 0)
        self._stage = None
        self._fill = 0
        # Zero-copy gathering: pushes that are contiguous fp16 device batches of a multiple of 256 images are not copied
        # into a staging batch but remembered as 256-image blocks, and the tower reads them where they lie
        # (``lla_vit_b32_forward_gather``: 256 images are 49 whole 256-row tiles of the patch-embedding GEMM, the only
        # kernel that reads the images).  The copy is 301 KB per image read and written again -- 2 % of a tower pass.
        self._blocks = []          # (tensor view of <= 256 images) of the pass being gathered
        self.gathered_passes = 0   # tower passes that read donated batches in place (tests look at it)
        self._block_images = 0
        self._gather_ok = bool(getattr(compressor.clip, "forward_gather", None))
        self._free_stages = []     # staging batches whose group has been fetched: recycled, not re-allocated
        self._busy_stages = []     # ... of the group being filled (the lanes read them until it is fetched)
        self.zbufs = [None, None]
        self.cur = 0
        self.rows = 0
        self.pushes = 0
        self.out = []
        self.deferred = bool(getattr(compressor.clip, "supports_deferred", False))
        self._inflight = []
        self._pending = None      # (payload, pinned total, event, input refs) of the group being coded
        self._coder = None

    @torch.no_grad()
    def push(self, x, donate=False):
        """Queue a batch of images.  ``donate=False`` (default): the batch is the CALLER's -- everything the stream
        needs from it has been copied (in stream order) when ``push`` returns, so a preallocated buffer may be
        refilled for the next push.  ``donate=True``: the stream may keep VIEWS of a contiguous fp16 device batch of a
        multiple of 256 images and let the tower read it where it lies up to a whole pass (``coalesce`` images) later,
        at the next flush or at ``finish()`` -- the caller must not write to it (nor expect its memory back) until
        ``finish()`` has returned.  ``compress_dataset`` donates the batches it makes itself; same bytes either way."""
        c = self.c
        c._check_gpu()
        if not x.is_cuda:
            x = x.to(c.device)
        if isinstance(x, RaggedImages) or x.dtype == torch.uint8:  # raw RGB: resize / crop / normalise on the GPU
            x = c.preprocess_gpu(x)
        B = x.shape[0]
        if B == 0:
            return
        if donate and self.coalesce and self._gatherable(x) and not self._fill:
            self._gather_in(x)
            return
        if self._blocks:               # a batch the tower cannot read in place: what was gathered so far goes first
            self._flush_blocks()
        if self.coalesce and (B < self._target() or self._fill):
            self._stage_in(x)
            return
        if self._ramp:      # a batch of at least the current pass size goes as it is; the ramp moves past it
            self._ramp = [r for r in self._ramp if r > B]
        self._run_tower(x)

    def _gatherable(self, x):
        """Can the tower read this push in place, as part of a pass of `coalesce` images?"""
        if not self._gather_ok or self._ramp or isinstance(x, RaggedImages):
            return False
        chunk = int(getattr(self.c.clip, "chunk", 0) or 0)
        # (a pass is at most 64 pieces -- GemmParams::a_chunk -- and one library slice: csrc/switches_product.cpp default_chunk)
        return (x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and x.dim() == 4 and x.shape[0] % 256 == 0
                and self.coalesce % 256 == 0 and 256 <= self.coalesce <= min(_LIB_SLICE, 64 * 256) and chunk <= 0
                and (not self._blocks or (self._blocks[0].shape[1:] == x.shape[1:] and self._blocks[0].device == x.device)))

    def _gather_in(self, x):
        for i in range(0, x.shape[0], 256):
            self._blocks.append(x[i:i + 256])
            self._block_images += 256
            if self._block_images == self.coalesce:
                self._flush_blocks()

    def _flush_blocks(self):
        if self._blocks:
            blocks, n = self._blocks, self._block_images
            self._blocks, self._block_images = [], 0
            self.gathered_passes += 1
            self._run_tower(None, blocks=blocks, B=n)

    def _target(self):
        """Images the staging batch gathers before the next tower pass."""
        return self._ramp[0] if self._ramp else self.coalesce

    def _stage_in(self, x):
        """Append x to the staging batch; run the tower whenever the staging batch is full."""
        pos, B = 0, x.shape[0]
        while pos < B:
            if self._stage is None or self._stage.shape[1:] != x.shape[1:] or self._stage.device != x.device:
                self._flush_stage()
                want = (self.coalesce,) + tuple(x.shape[1:])
                self._free_stages = [t for t in self._free_stages if tuple(t.shape) == want and t.device == x.device]
                self._stage = (self._free_stages.pop() if self._free_stages else
                               torch.empty(want, dtype=torch.float16, device=x.device))
            n = min(B - pos, self._target() - self._fill)
            self._stage[self._fill:self._fill + n].copy_(x[pos:pos + n])     # (converts to fp16 on the way)
            self._fill += n
            pos += n
            if self._fill == self._target():
                if self._ramp:
                    self._ramp.pop(0)
                self._flush_stage()

    def _flush_stage(self):
        if self._stage is not None and self._fill:
            stage, n = self._stage, self._fill
            # the lanes read it until its group has been fetched: another buffer next time (peak: the stages of
            # two groups, i.e. 2 * group * coalesce images of 301 KB -- 9.9 GB at the defaults)
            self._stage, self._fill = None, 0
            self._run_tower(stage[:n], owner=stage)

    @torch.no_grad()
    def _run_tower(self, x, owner=None, blocks=None, B=None):
        c = self.c
        if blocks is None:
            B = x.shape[0]
            # The lanes read the batch after this call returns (deferred passes), outside the current stream's
            # order: convert HERE so that the tensor kept in `_inflight` is the one they read -- a converted
            # temporary made further down would go back to the caching allocator while still being read.
            if x.dtype != torch.float16 or not x.is_contiguous():
                x = x.half().contiguous()
        dev = blocks[0].device if blocks is not None else x.device
        zb = self.zbufs[self.cur]
        if zb is not None and self.rows + B > zb.shape[0]:
            self._encode()
            zb = self.zbufs[self.cur]
        if zb is None or B > zb.shape[0]:     # (rows == 0 here; the other buffer may still be read by the coder)
            zb = self.zbufs[self.cur] = torch.empty((self.group * 1024 + B, c.z_dim), dtype=torch.float16,
                                                    device=dev)
        if owner is not None:
            # the staging batch belongs to the group whose tower pass reads it -- registered only now, AFTER the
            # overflow _encode() above, which hands the busy list to the PREVIOUS group (it would be recycled when
            # that group's coding is done, i.e. possibly while this pass still reads it)
            self._busy_stages.append(owner)
        if blocks is not None:
            c.clip.forward_gather(blocks, B, out=zb[self.rows:self.rows + B])   # ... and reads its images where they lie
            self._inflight.append(blocks)      # (views: they keep the pushed batches alive until the group is fetched)
        elif self.deferred:
            c.clip(x, out=zb[self.rows:self.rows + B], deferred=True)
            self._inflight.append(x)
        else:
            c.clip(x, out=zb[self.rows:self.rows + B])   # the tower writes its rows in place
            self._inflight.append(x)
        self.rows += B
        self.pushes += 1
        if self.rows >= self.group * 1024:      # `group` counts thousands of images, whatever the tower batch size
            self._encode()

    def _collect(self):
        """Fetch the bytes of the group whose coding was queued by the previous _encode."""
        if self._pending is None:
            return
        payload, total_host, done, refs = self._pending
        done.synchronize()
        self._free_stages += refs[2]             # nothing reads this group's staging batches any more
        total = int(total_host[0])
        with torch.cuda.stream(self._coder):
            if self.on_device:      # (a copy of the used bytes: `payload` is sized for the worst case, 3.3 KB per image)
                self.out.append(payload[:total].clone())
            else:
                self.out.append(payload[:total].cpu().numpy())
        self._pending = None

    @torch.no_grad()
    def _encode(self):
        self._collect()
        if self.rows:
            c = self.c
            zb = self.zbufs[self.cur]
            if self._coder is None:
                self._coder = torch.cuda.Stream(device=zb.device)
            self._coder.wait_stream(torch.cuda.current_stream(zb.device))
            with torch.cuda.stream(self._coder):
                if self.deferred:
                    c.clip.join(zb.device)          # the coder stream (current here) waits for both lanes
                payload, offsets, _ = c.entropy_bottleneck.encode_device(
                    zb[:self.rows], c._tables(), record_prefix=True)
                total_host = torch.empty(1, dtype=offsets.dtype, pin_memory=True)
                total_host.copy_(offsets[-1:], non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._coder)
            self._pending = (payload, total_host, done, (self._inflight, zb, self._busy_stages))
            self._inflight, self._busy_stages = [], []
            self.cur ^= 1
        self.rows = 0
        self.pushes = 0

    def finish(self):
        """Code what is parked and return all record bytes pushed so far: a host uint8 array, or -- for a stream made
        with ``on_device=True`` -- a 1-D uint8 DEVICE tensor (the records never cross the bus: what a rank > 0 hands
        to the RCCL gather)."""
        self._flush_blocks()
        self._flush_stage()
        self._encode()
        self._collect()
        if self.on_device:
            if self._coder is not None:
                cur = torch.cuda.current_stream(self._coder.device)
                cur.wait_stream(self._coder)
                # the clones were allocated on the coder stream and are read by the cat on the current one: tell the
                # caching allocator, or it may hand their blocks back to the coder stream while the cat is in flight
                # (a caller that calls finish() now and then and keeps pushing)
                for t in self.out:
                    t.record_stream(cur)
            body = torch.cat(self.out) if self.out else torch.zeros(0, dtype=torch.uint8, device=self.c.device)
        else:
            body = np.concatenate(self.out) if self.out else np.zeros(0, np.uint8)
        self.out = []
        return body


class SyntheticImages:
    """N CLIP-normalised fp16 NHWC images generated on the device batch by batch, never
    materialised (BASELINE.json configs[3]: 1M x 224 x 224 x 3 would be 301 GB).  Image i is
    a pure function of (seed, i): u8 ~ U{0..255} from a counter-based hash, so any sharding of
    the index range produces the same pixels and hence the same file."""

    def __init__(self, n, seed=0):
        self.n, self.seed = int(n), int(seed)

    def __len__(self):
        return self.n

    def device_batch(self, lo, hi, device):
        """Images lo .. hi-1 as one fp16 NHWC device batch (``lla_synthetic_images``: one HBM-write-bound
        kernel; the formula is in include/lossyless_amd.h and, as torch ops, in ``reference_batch``)."""
        from .preprocess import CLIP_MEAN, CLIP_STD
        dev = torch.device(device)
        out = torch.empty((hi - lo, 224, 224, 3), dtype=torch.float16, device=dev)
        _lib.require_cuda(out, "SyntheticImages batches")
        mean, std = (ctypes.c_float * 3)(*CLIP_MEAN), (ctypes.c_float * 3)(*CLIP_STD)
        with torch.cuda.device(dev):
            rc = _lib.lib().lla_synthetic_images(self.seed, lo, hi - lo, mean, std, _lib.ptr(out),
                                                 _lib.stream_ptr(dev))
        _lib.check(rc, "lla_synthetic_images")
        return out

    def reference_batch(self, lo, hi, device="cpu"):
        """The same images from the defining formula in plain torch int64 / fp32 ops (any device; what the
        kernel is tested against)."""
        from .preprocess import CLIP_MEAN, CLIP_STD
        per = 224 * 224 * 3
        idx = torch.arange(lo * per, hi * per, device=device, dtype=torch.int64)
        h = (idx ^ (self.seed * 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF)) * 0x2545F4914F6CDD1D
        h = h ^ (h >> 29)
        h = h * 0x94D049BB133111EB
        u8 = ((h >> 40) & 0xFF).to(torch.float32).reshape(hi - lo, 224, 224, 3)
        mean = torch.tensor(CLIP_MEAN, device=device)
        std = torch.tensor(CLIP_STD, device=device)
        return ((u8 / 255 - mean) / std).half()


# Container field helpers under the reference's names (hub/compressor.py:258-275): unsigned
# 32-bit big-endian integers and raw byte runs.  `fmt` is kept for signature compatibility.
def write_uints(fd, values, fmt=">{:d}I"):
    fd.write(b"".join(int(v).to_bytes(4, "big") for v in values))


def write_bytes(fd, values, fmt=">{:d}s"):
    if values:
        fd.write(bytes(values))


def read_uints(fd, n, fmt=">{:d}I"):
    raw = fd.read(4 * n)
    return tuple(int.from_bytes(raw[4 * i:4 * i + 4], "big") for i in range(n))


def read_bytes(fd, n, fmt=">{:d}s"):
    return fd.read(n)
