"""ctypes binding of liblossyless_amd.so -- the C-ABI declared in include/lossyless_amd.h.

There is no CPU fallback: if the HIP library is missing, loading raises, and every
device entry point raises if it returns a non-zero status.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: it brings the HIP runtime the .so binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LLA_LIB names another build of the same ABI (tools/ load the -DLLA_ABLATION build this way)
LIB_PATH = os.environ.get("LLA_LIB") or os.path.join(_HERE, "liblossyless_amd.so")

LLA_OK = 0
ABI_VERSION = 4
LLA_Z_F16, LLA_Z_F32 = 1, 2
LLA_LAYOUT_NHWC, LLA_LAYOUT_NCHW = 0, 1
LLA_EPI_F16, LLA_EPI_QUICKGELU_F16, LLA_EPI_RESID_F32, LLA_EPI_RELU_F16, LLA_EPI_ADD_RELU_F16 = 0, 1, 2, 4, 5
_ERR = {-1: "LLA_EINVAL", -2: "LLA_ECAP", -3: "LLA_EHIP", -4: "LLA_EDATA"}

# enum lla_vit_param
VIT_GLOBAL = dict(CONV1_NHWC=0, CONV1_NCHW=1, CLASS_EMB=2, POS_EMB=3, LN_PRE_W=4, LN_PRE_B=5,
                  LN_POST_W=6, LN_POST_B=7, PROJ_T=8)
VIT_LAYER = dict(LN1_W=16, LN1_B=17, QKV_W=18, QKV_B=19, OUT_W=20, OUT_B=21, LN2_W=22, LN2_B=23,
                 FC_W=24, FC_B=25, CPROJ_W=26, CPROJ_B=27)

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
_SIGNATURES = {
    "lla_abi_version": (_i, []),
    "lla_last_hip_error": (_i, []),
    "lla_source_sha": (ctypes.c_char_p, []),
    "lla_pmf_to_quantized_cdf": (_i, [_vp, _i, _i, _vp]),
    "lla_rans_max_encoded_bytes": (_sz, [_i]),
    "lla_container_index": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "lla_rans_encode_batch_host": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "lla_rans_decode_batch_host": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "lla_dequantise_host": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "lla_quantise": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lla_rans_encode_batch": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "lla_quantise_encode": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz,
                                 _vp, _vp, _vp]),
    "lla_rans_compact_workspace_bytes": (_sz, [_i]),
    "lla_rans_compact": (_i, [_vp, _sz, _vp, _i, _i, _vp, _sz, _vp, _vp, _sz, _vp]),
    "lla_rans_decode_batch": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "lla_rans_encode_indexed": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp, _vp]),
    "lla_rans_decode_indexed": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lla_dequantise": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lla_represent": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "lla_pillow_bicubic_ksize": (_i, [_i, _i]),
    "lla_pillow_bicubic_taps": (_i, [_i, _i, _i, _i, _i, _vp, _vp]),
    "lla_preprocess_ragged_lds_bytes": (_sz, [_vp, _i, _vp, _i, _i]),
    "lla_preprocess_clip_ragged": (_i, [_vp, _i, _i, _sz, _vp, _vp, _vp, _vp]),
    "lla_synthetic_images": (_i, [ctypes.c_uint64, ctypes.c_uint64, _i, _vp, _vp, _vp, _vp]),
    "lla_preprocess_workspace_bytes": (_sz, [_i, _i]),
    "lla_preprocess_clip": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _sz,
                                 _vp, _vp]),
    "lla_vit_b32_weights_bytes": (_sz, []),
    "lla_vit_b32_param_offset": (_sz, [_i, _i]),
    "lla_vit_b32_param_bytes": (_sz, [_i]),
    "lla_vit_b32_workspace_bytes": (_sz, [_i]),
    "lla_vit_b32_forward": (_i, [_vp, _i, _i, _vp, _vp, _sz, _i, _vp, _vp]),
    "lla_profiler_create": (_i, [ctypes.POINTER(ctypes.c_void_p), _i]),
    "lla_profiler_destroy": (_i, [_vp]),
    "lla_profiler_collect": (_i, [_vp, _vp, _vp, _vp]),
    "lla_vit_b32_forward_profiled": (_i, [_vp, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    "lla_tower_create": (_i, [ctypes.POINTER(ctypes.c_void_p)]),
    "lla_tower_destroy": (_i, [_vp]),
    "lla_tower_join": (_i, [_vp, _vp]),
    "lla_tower_set_option": (_i, [_vp, _i, _i]),
    "lla_vit_b32_forward_lanes": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _i, _vp, _vp, _i]),
    "lla_vit_b32_forward_gather": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "lla_gemm_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lla_patch_embed_f16": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "lla_gemm_f16_ex": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "lla_gemm_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lla_conv3x3_relu_f16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "lla_conv3x3_direct_relu_f16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "lla_conv3x3_rgb_s2_relu_f16": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp]),
    "lla_rn50_bottleneck_f16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "lla_rn50_weights_bytes": (_sz, []),
    "lla_rn50_conv_count": (_i, []),
    "lla_rn50_conv_desc": (_i, [_i, _vp]),
    "lla_rn50_fused_desc": (_i, [_i, _vp]),
    "lla_rn50_attnpool_offsets": (_i, [_vp]),
    "lla_rn50_workspace_bytes": (_sz, [_i]),
    "lla_rn50_forward": (_i, [_vp, _i, _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    "lla_layernorm768": (_i, [_vp, _sz, _vp, _vp, _vp, _i, _vp]),
    "lla_attention50": (_i, [_vp, _vp, _i, _vp]),
}
EXPORTS = tuple(sorted(_SIGNATURES))

_lib = None
_CSRC = os.path.join(_HERE, "csrc")


def tree_sha():
    """sha of the kernel sources in this tree (csrc/source_sha.py: what the Makefile compiles into the library), or
    None when the sources are not there (an installed copy of the package with a prebuilt library)."""
    script = os.path.join(_CSRC, "source_sha.py")
    if not os.path.exists(script):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("_lla_source_sha", script)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_sha(_CSRC)


def built_sha(path=None):
    """sha compiled into the library FILE at `path` (read from its bytes: nothing is loaded), or None."""
    import re
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = re.search(rb"LLA_SOURCE_SHA=([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else None


def stale(path=None):
    """Why the library at `path` must be (re)built -- missing, or built from other sources than this tree's -- or None."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return f"{path} is missing"
    want, have = tree_sha(), built_sha(path)
    if want is not None and have != want:
        return f"{path} was built from sources {have}, the tree is {want}"
    return None


def ensure_built(path=None):
    """(Re)build the library at `path` (the product library, the -DLLA_ABLATION build or a `make variant` build is told
    from its file name) if it is missing or stale.  Call BEFORE the first lib(): a loaded library cannot be replaced."""
    import subprocess
    path = path or LIB_PATH
    why = stale(path)
    if why is None:
        return False
    name = os.path.basename(path)
    if os.path.dirname(os.path.abspath(path)) == os.path.join(_HERE, "variants"):
        raise RuntimeError(f"{why}: rebuild it with the DEFS it was made with (make -C lossyless_amd/csrc variant NAME=... DEFS=...)")
    target = {"liblossyless_amd.so": [], "liblossyless_amd_ablation.so": ["ablation"], "liblossyless_amd_probes.so": ["probes"]}.get(name)
    if target is None or _lib is not None:
        raise RuntimeError(why)
    subprocess.check_call(["make", "-j8", "-C", _CSRC, *target])
    return True


def lib():
    """Load (once) and return the ctypes library; raises if it has not been built, or was built from other sources than
    the tree's (a stale binary must not pass for the code under test)."""
    global _lib
    if _lib is None:
        why = stale(LIB_PATH)
        if why is not None:
            raise RuntimeError(
                f"{why}: the HIP extension has not been (re)built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "lossyless_amd/csrc`). lossyless_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if L.lla_abi_version() != ABI_VERSION:
            raise RuntimeError("liblossyless_amd.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what):
    if rc != LLA_OK:
        extra = f" (hipError {lib().lla_last_hip_error()})" if rc == -3 else ""
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}{extra}")


def ptr(t):
    """Raw address of a tensor / None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: lossyless_amd runs on MI355X only "
                           "(no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


class Tower:
    """Owner of one ``lla_tower_create`` handle (the two tower lanes of one device)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().lla_tower_create(ctypes.byref(self.handle)), "lla_tower_create")

    OPT_LNX, OPT_LNX_WAIT = 1, 2      # LLA_TOWER_OPT_* (include/lossyless_amd.h)

    def set_option(self, option, value):
        """``lla_tower_set_option``: choose between code paths that give the same embeddings bit for bit (tests)."""
        check(lib().lla_tower_set_option(self.handle, int(option), int(value)), "lla_tower_set_option")

    def join(self):
        """The current stream waits for everything queued on the lanes."""
        check(lib().lla_tower_join(self.handle, stream_ptr(self.device)), "lla_tower_join")

    def close(self):
        if self.handle:
            lib().lla_tower_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass


def upload_in_pieces(module, name, fn, step=32 << 20):
    """Inside ``nn.Module._apply(fn)``: if ``fn`` moves the 1-D CPU buffer ``name`` to a GPU, move it there in
    ``step``-byte pieces instead (see ``VisionTransformer._apply``: a pageable source of >= 128 MB would be pinned
    in place by the HIP runtime and left registered)."""
    blob = module._buffers.get(name)
    if blob is None or blob.device.type != "cpu" or blob.dim() != 1:
        return
    like = fn(blob[:0])
    if like.device.type != "cuda" or like.dtype != blob.dtype:
        return
    up = torch.empty(blob.shape, dtype=blob.dtype, device=like.device)
    n = max(step // blob.element_size(), 1)
    for i in range(0, blob.numel(), n):
        up[i:i + n].copy_(blob[i:i + n])
    module._buffers[name] = up
