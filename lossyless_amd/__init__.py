"""lossyless_amd -- MI355X-native ``compress_dataset`` hot path of YannDubs/lossyless.

CLIP ViT-B/32 image encoder -> factorized EntropyBottleneck -> rANS -> ``.bin``, behind
the reference's ``ClipCompressor`` API (hub/compressor.py), running in hand-written HIP
kernels for gfx950 reached through the C ABI in ``include/lossyless_amd.h``.
"""
import os as _os

# Host-runtime setting, made before the HIP runtime starts (importing torch does not start it; an explicit setting
# by the user wins; too late = no effect).  By default ROCr backs pinned host memory with anonymous pages + a KFD
# userptr registration.  Every fork() -- each DataLoader worker of the reference call, hub/compressor.py:154,186 --
# write-protects those pages, the kernel driver evicts the process's GPU queues and re-pins everything before
# they run again: measured on the MI355X host, the first H2D copy after 4 forks with 1 GiB pinned waits 14.6 s
# (tools/h2d_probe.py), and a 16-worker compress_dataset spent 2-15 s of every call there.  With "0" pinned memory
# is a GTT buffer object the children do not share: 1.4 ms, same 56 GB/s over the bus.
# Both are process-wide and change H2D copy behaviour for every HIP user in the process: a value already in the
# environment wins, and LOSSYLESS_AMD_KEEP_HOST_RUNTIME_DEFAULTS=1 leaves both alone.
_keep = _os.environ.get("LOSSYLESS_AMD_KEEP_HOST_RUNTIME_DEFAULTS", "0") == "1"
if not _keep:
    _os.environ.setdefault("HSA_USERPTR_FOR_PAGED_MEM", "0")
# Same mechanism, second source: the HIP runtime pins a PAGEABLE source of >= 128 MiB in place for an H2D copy and
# keeps that userptr registration for reuse (``model.cuda()``, ``images.cuda()``); 4 forks after one 175 MB upload
# stalled the GPU for 3.4 s.  Raise the threshold (MiB) so that pageable copies always go through the runtime's
# staging buffers.
if not _keep:
    _os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "65536")

from .compressor import ClipCompressor  # noqa: F401
from .entropy import EntropyBottleneck  # noqa: F401
from .clip_vit import VisionTransformer, synthetic_vit_state_dict  # noqa: F401

__all__ = ["ClipCompressor", "EntropyBottleneck", "VisionTransformer", "synthetic_vit_state_dict"]
