"""CLIP ViT-B/32 visual tower on MI355X: host side of ``lla_vit_b32_forward``.

Stands in for ``clip.load("ViT-B/32")[0].visual`` as used at hub/compressor.py:39-40,93
(``clip==1.0``, not vendored; architecture in SURVEY.md section 9.3).  Weights are taken in
the OpenAI state-dict layout (keys under ``visual.`` with the prefix stripped), packed once
into the device blob described by ``enum lla_vit_param`` in include/lossyless_amd.h, and the
whole forward pass is one C-ABI call -- no torch ops on the data path.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib

WIDTH, LAYERS, HEADS, PATCH, RES, OUT, TOKENS, MLP = 768, 12, 12, 32, 224, 512, 50, 3072

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # lossyless/helpers.py:252
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)   # lossyless/helpers.py:260


def synthetic_vit_state_dict(seed=1):
    """Random-init ViT-B/32 visual weights in the OpenAI layout:
    N(0, 0.02^2) matrices and conv, LayerNorm gamma=1 beta=0, class / positional / proj
    ~ N(0, 1/768).  Used when no real CLIP weights are available (there is no network)."""
    g = torch.Generator().manual_seed(seed)
    n = lambda *s, std: torch.randn(*s, generator=g) * std
    sd = {
        "conv1.weight": n(WIDTH, 3, PATCH, PATCH, std=0.02),
        "class_embedding": n(WIDTH, std=WIDTH ** -0.5),
        "positional_embedding": n(TOKENS, WIDTH, std=WIDTH ** -0.5),
        "ln_pre.weight": torch.ones(WIDTH), "ln_pre.bias": torch.zeros(WIDTH),
        "ln_post.weight": torch.ones(WIDTH), "ln_post.bias": torch.zeros(WIDTH),
        "proj": n(WIDTH, OUT, std=WIDTH ** -0.5),
    }
    for l in range(LAYERS):
        p = f"transformer.resblocks.{l}."
        sd[p + "ln_1.weight"] = torch.ones(WIDTH)
        sd[p + "ln_1.bias"] = torch.zeros(WIDTH)
        sd[p + "attn.in_proj_weight"] = n(3 * WIDTH, WIDTH, std=0.02)
        sd[p + "attn.in_proj_bias"] = n(3 * WIDTH, std=0.02)
        sd[p + "attn.out_proj.weight"] = n(WIDTH, WIDTH, std=0.02)
        sd[p + "attn.out_proj.bias"] = n(WIDTH, std=0.02)
        sd[p + "ln_2.weight"] = torch.ones(WIDTH)
        sd[p + "ln_2.bias"] = torch.zeros(WIDTH)
        sd[p + "mlp.c_fc.weight"] = n(MLP, WIDTH, std=0.02)
        sd[p + "mlp.c_fc.bias"] = n(MLP, std=0.02)
        sd[p + "mlp.c_proj.weight"] = n(WIDTH, MLP, std=0.02)
        sd[p + "mlp.c_proj.bias"] = n(WIDTH, std=0.02)
    return sd


def clip_like_vit_state_dict(seed=1, logit_gain=10.0, massive=30.0, ln_gain=20.0):
    """Random-init weights with the STATISTICS that make a trained CLIP ViT-B/32 hard for fp16 activations
    (the real checkpoint cannot be fetched offline): LayerNorm gains spread around 1 with four channels at
    ``ln_gain`` x, "massive activations" (two class-token channels and two every-token channels of the
    residual stream carrying ``massive`` from the first blocks on), and per-head query gains so that the
    attention logits have a standard deviation of a few units -- softmax rows peak at 0.3-0.9 instead of the
    flat 0.03 of N(0, 0.02^2) weights.  Same layout as ``synthetic_vit_state_dict``."""
    sd = synthetic_vit_state_dict(seed)
    g = torch.Generator().manual_seed(seed + 1000)
    hot = [41, 133, 361, 682]
    for k in list(sd):
        if k.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight")):
            sd[k] = (1 + 0.25 * torch.randn(WIDTH, generator=g)).abs()
            sd[k][hot] *= ln_gain if "ln_pre" in k else 1.0
        elif k.endswith(("ln_1.bias", "ln_2.bias", "ln_pre.bias", "ln_post.bias", "c_fc.bias",
                         "out_proj.bias")):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    sd["class_embedding"][[5, 300]] += massive                     # class-token outliers
    sd["transformer.resblocks.1.mlp.c_proj.bias"] = sd["transformer.resblocks.1.mlp.c_proj.bias"].clone()
    sd["transformer.resblocks.1.mlp.c_proj.bias"][[77, 500]] += massive   # outliers on every token from block 1 on
    hd = WIDTH // HEADS
    for l in range(LAYERS):
        w = sd[f"transformer.resblocks.{l}.attn.in_proj_weight"]
        b = sd[f"transformer.resblocks.{l}.attn.in_proj_bias"]
        gains = logit_gain * (0.5 + torch.rand(HEADS, generator=g))           # per-head query gain
        for h in range(HEADS):
            w[h * hd:(h + 1) * hd] *= gains[h]
            b[h * hd:(h + 1) * hd] *= gains[h]
    return sd


def load_clip_visual_state_dict(path):
    """Read OpenAI ``ViT-B-32.pt`` (TorchScript archive) or a plain state-dict and return
    the visual tower's tensors with the ``visual.`` prefix stripped."""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except Exception:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if "state_dict" in sd:
            sd = sd["state_dict"]
    if any(k.startswith("visual.") for k in sd):
        sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    return sd


def pack_weights(sd):
    """OpenAI-layout state-dict -> uint8 numpy blob in the library's layout.  Matrices
    become fp16 (what ``clip.load`` keeps on a GPU); vectors are rounded to fp16 and
    stored as fp32 (CLIP's LayerNorm upcasts fp16 parameters)."""
    L = _lib.lib()
    blob = np.zeros(int(L.lla_vit_b32_weights_bytes()), dtype=np.uint8)

    def put(pid, layer, t, half, exact=False):
        t = t.detach().to("cpu", torch.float32).contiguous()
        arr = t.half().numpy() if half else (t.numpy() if exact else t.half().float().numpy())
        raw = arr.reshape(-1).view(np.uint8)
        off = int(L.lla_vit_b32_param_offset(pid, layer))
        nbytes = int(L.lla_vit_b32_param_bytes(pid))
        if raw.nbytes != nbytes:
            raise ValueError(f"param {pid}: {raw.nbytes} bytes, library expects {nbytes}")
        blob[off:off + nbytes] = raw

    G, Y = _lib.VIT_GLOBAL, _lib.VIT_LAYER
    conv = sd["conv1.weight"]                                     # [768, 3, 32, 32]
    put(G["CONV1_NCHW"], 0, conv.reshape(WIDTH, -1), True)        # K order (c, kh, kw)
    put(G["CONV1_NHWC"], 0, conv.permute(0, 2, 3, 1).reshape(WIDTH, -1), True)  # (kh, kw, c)
    put(G["CLASS_EMB"], 0, sd["class_embedding"], False)
    put(G["POS_EMB"], 0, sd["positional_embedding"], False)
    put(G["LN_PRE_W"], 0, sd["ln_pre.weight"], False)
    put(G["LN_PRE_B"], 0, sd["ln_pre.bias"], False)
    put(G["LN_POST_W"], 0, sd["ln_post.weight"], False)
    put(G["LN_POST_B"], 0, sd["ln_post.bias"], False)
    put(G["PROJ_T"], 0, sd["proj"].t(), True)
    for l in range(LAYERS):
        p = f"transformer.resblocks.{l}."
        put(Y["LN1_W"], l, sd[p + "ln_1.weight"], False)
        put(Y["LN1_B"], l, sd[p + "ln_1.bias"], False)
        put(Y["QKV_W"], l, sd[p + "attn.in_proj_weight"], True)
        put(Y["QKV_B"], l, sd[p + "attn.in_proj_bias"], False)
        put(Y["OUT_W"], l, sd[p + "attn.out_proj.weight"], True)
        put(Y["OUT_B"], l, sd[p + "attn.out_proj.bias"], False)
        put(Y["LN2_W"], l, sd[p + "ln_2.weight"], False)
        put(Y["LN2_B"], l, sd[p + "ln_2.bias"], False)
        put(Y["FC_W"], l, sd[p + "mlp.c_fc.weight"], True)
        put(Y["FC_B"], l, sd[p + "mlp.c_fc.bias"], False)
        put(Y["CPROJ_W"], l, sd[p + "mlp.c_proj.weight"], True)
        put(Y["CPROJ_B"], l, sd[p + "mlp.c_proj.bias"], False)
    return blob


class VisionTransformer(nn.Module):
    """``model.visual`` replacement: ``forward(X) -> z`` with X [B,3,224,224] fp16 (NCHW, what
    the reference feeds, hub/compressor.py:187) or [B,224,224,3] fp16 (NHWC fast path)."""

    def __init__(self, state_dict, chunk=0):
        super().__init__()
        self.register_buffer("blob", torch.from_numpy(pack_weights(state_dict)), persistent=False)
        self.chunk = int(chunk)
        self._ws = None
        self._tower = None
        self.input_resolution = RES
        self.output_dim = OUT

    def __getstate__(self):
        """Pickling / ``copy.deepcopy`` / ``torch.save`` of a module that has run: the tower handle (a ctypes
        pointer owned by THIS object) and the workspace are per-process scratch, re-made on first use."""
        state = self.__dict__.copy()
        state["_ws"] = None
        state["_tower"] = None
        return state

    def _apply(self, fn, recurse=True):
        """``.to(device)`` / ``.cuda()``: the packed weights (175 MB) go up in 32 MB pieces.  The HIP runtime pins
        a pageable source of >= 128 MB IN PLACE for the copy (a userptr buffer object it keeps for reuse); every
        later fork() -- a DataLoader worker -- write-protects those pages and the driver holds the process's GPU
        queues until it has re-pinned them: 4 forks cost 3.4 s of GPU stall after one whole-blob upload, 6 ms after
        a piecewise one (tools/fork_probe3.py)."""
        _lib.upload_in_pieces(self, "blob", fn)
        return super()._apply(fn, recurse)

    def _workspace(self, dev):
        need = int(_lib.lib().lla_vit_b32_workspace_bytes(self.chunk))  # <= 0: library default
        if self._ws is None or self._ws.device != dev or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def tower(self, dev):
        """This module's tower handle (the two lanes) on ``dev``, created on first use."""
        dev = torch.device(dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if self._tower is None or self._tower.device != dev:
            if self._tower is not None:
                self._tower.close()
            self._tower = _lib.Tower(dev)
        return self._tower

    @staticmethod
    def layout_of(X):
        if X.dim() != 4:
            raise ValueError("expected a 4-D image batch")
        if tuple(X.shape[1:]) == (3, RES, RES):
            return _lib.LLA_LAYOUT_NCHW
        if tuple(X.shape[1:]) == (RES, RES, 3):
            return _lib.LLA_LAYOUT_NHWC
        raise ValueError(f"expected [B,3,{RES},{RES}] or [B,{RES},{RES},3], got {tuple(X.shape)}")

    supports_deferred = True

    def forward(self, X, out=None, profiler=None, deferred=False):
        """``deferred=True`` (streaming callers): the pass is queued on the library's two tower lanes and
        NOT joined back into the current stream -- ``X`` must stay alive and ``out`` unread until
        ``join()``.  Bit-identical embeddings either way."""
        layout = self.layout_of(X)
        if deferred and (X.dtype != torch.float16 or not X.is_contiguous()):
            # a converted copy would die with this call while the lanes still read it
            raise ValueError("deferred passes read X after the call returns: pass a contiguous fp16 tensor "
                             "and keep it alive until join()")
        if X.dtype != torch.float16:
            X = X.half()
        X = X.contiguous()
        _lib.require_cuda(X, "X")
        if self.blob.device != X.device:
            raise RuntimeError("weights and input are on different devices")
        B = X.shape[0]
        L = _lib.lib()
        ws = self._workspace(X.device)
        z = out if out is not None else torch.empty((B, OUT), dtype=torch.float16, device=X.device)
        with torch.cuda.device(X.device):
            tower = self.tower(X.device)
            if profiler is None:
                rc = L.lla_vit_b32_forward_lanes(
                    tower.handle, _lib.ptr(X), layout, B, _lib.ptr(self.blob), _lib.ptr(ws), ws.numel(),
                    self.chunk, _lib.ptr(z), _lib.stream_ptr(X.device), 1 if deferred else 0)
            else:
                tower.join()     # the profiled pass runs on the current stream through lane 0's slice buffers
                rc = L.lla_vit_b32_forward_profiled(
                    _lib.ptr(X), layout, B, _lib.ptr(self.blob), _lib.ptr(ws), ws.numel(), self.chunk,
                    _lib.ptr(z), _lib.stream_ptr(X.device), profiler.handle)
        _lib.check(rc, "lla_vit_b32_forward")
        return z

    def forward_gather(self, blocks, B, out=None):
        """The pass over a batch that lies in pieces: ``blocks[i]`` is a contiguous fp16 CUDA tensor view holding images
        256 i .. 256 i + 255 of the batch (the last one B - 256 (len - 1) of them), all in one layout
        (``lla_vit_b32_forward_gather``: nothing is copied together first).  The caller keeps the tensors alive until the
        pass has run.  Same embeddings as ``forward(torch.cat(blocks))``."""
        import ctypes
        layout = self.layout_of(blocks[0])
        dev = blocks[0].device
        for t in blocks:
            if t.dtype != torch.float16 or not t.is_contiguous() or t.device != dev or self.layout_of(t) != layout:
                raise ValueError("forward_gather takes contiguous fp16 blocks of one layout on one device")
            _lib.require_cuda(t, "block")
        if self.blob.device != dev:
            raise RuntimeError("weights and input are on different devices")
        L = _lib.lib()
        ws = self._workspace(dev)
        z = out if out is not None else torch.empty((B, OUT), dtype=torch.float16, device=dev)
        ptrs = (ctypes.c_void_p * len(blocks))(*[t.data_ptr() for t in blocks])
        with torch.cuda.device(dev):
            rc = L.lla_vit_b32_forward_gather(self.tower(dev).handle, ptrs, len(blocks), 256, layout, B,
                                              _lib.ptr(self.blob), _lib.ptr(ws), ws.numel(), _lib.ptr(z),
                                              _lib.stream_ptr(dev))
        _lib.check(rc, "lla_vit_b32_forward_gather")
        return z

    def join(self, device=None):
        """Make the current stream wait for every deferred pass queued so far."""
        dev = self.blob.device if device is None else device
        with torch.cuda.device(dev):
            self.tower(dev).join()


class KernelProfiler:
    """HIP-event timing of every kernel the tower launches, by class (``lla_profiler_*``)."""
    CLASSES = ("gemm", "layernorm", "attention")

    def __init__(self, max_launches=4096):
        import ctypes
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib().lla_profiler_create(ctypes.byref(self.handle), max_launches),
                   "lla_profiler_create")

    def collect(self):
        """-> {class: dict(ms=, work=, launches=)} accumulated since the last collect."""
        import ctypes
        n = len(self.CLASSES)
        ms = (ctypes.c_double * n)()
        work = (ctypes.c_double * n)()
        cnt = (ctypes.c_longlong * n)()
        _lib.check(_lib.lib().lla_profiler_collect(self.handle, ms, work, cnt), "lla_profiler_collect")
        return {c: dict(ms=ms[i], work=work[i], launches=int(cnt[i]))
                for i, c in enumerate(self.CLASSES)}

    def close(self):
        if self.handle:
            _lib.lib().lla_profiler_destroy(self.handle)
            self.handle = None


from .preprocess import ClipPreprocess  # noqa: E402,F401  (kept importable from here)


def resolve_clip_weights(spec=None):
    """``spec``: a state-dict, a path to OpenAI ``ViT-B-32.pt`` / a state-dict file, the literal
    ``"synthetic"`` (seed-1 random weights: tests, bench and smoke only), or None = take the
    path from ``$LOSSYLESS_CLIP_WEIGHTS``.  Returns (state_dict, description).

    The shipped rate models were trained on real CLIP features, so a compressor built on anything
    else produces meaningless rates: with nothing configured this raises instead of guessing
    (``clip.load`` -- what the reference calls at hub/compressor.py:39 -- needs the network)."""
    if isinstance(spec, dict):
        return spec, "state-dict"
    if spec is None:
        spec = os.environ.get("LOSSYLESS_CLIP_WEIGHTS")
        if not spec:
            # what the reference does (hub/compressor.py:39): clip.load("ViT-B/32") -- its download cache
            # first, then the clip package itself (which downloads when it can)
            cached = os.path.expanduser("~/.cache/clip/ViT-B-32.pt")
            if os.path.exists(cached):
                spec = cached
            else:
                try:
                    import clip  # noqa: PLC0415
                    model, _ = clip.load("ViT-B/32", device="cpu", jit=False)
                    return ({k: v for k, v in model.visual.state_dict().items()}, "clip.load('ViT-B/32')")
                except Exception as err:
                    raise ValueError(
                        "no CLIP ViT-B/32 weights found: pass clip_weights=<path to ViT-B-32.pt or a visual "
                        "state-dict>, set $LOSSYLESS_CLIP_WEIGHTS, or make `clip.load('ViT-B/32')` work "
                        f"(~/.cache/clip/ViT-B-32.pt is absent and clip.load failed: {err!r}). "
                        "clip_weights='synthetic' selects seed-1 random weights explicitly -- embeddings are "
                        "then NOT CLIP embeddings.") from None
    if spec == "synthetic":
        return synthetic_vit_state_dict(1), "synthetic-seed1"
    return load_clip_visual_state_dict(spec), str(spec)
