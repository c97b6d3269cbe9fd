"""CLIP RN50 visual tower on MI355X: host side of ``lla_rn50_forward``.

Stands in for ``clip.load("RN50")[0].visual`` as the reference's pretrained featuriser loads it
(lossyless/architectures.py:367-371, ``clip_rn50`` mode: 1024-dimensional output, :321,339) --
SURVEY.md 8(f) rank 4.  Weights come in the OpenAI state-dict layout (``visual.`` prefix stripped:
``conv1/bn1 .. conv3/bn3``, ``layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.1,downsample.2}``,
``attnpool.{positional_embedding,q_proj,k_proj,v_proj,c_proj}``); BatchNorm (eval mode, eps 1e-5) is
folded into the convolution weights / biases in float32 before the weights are rounded to fp16, and the
whole forward pass is one C-ABI call.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib

BLOCKS, PLANES = (3, 4, 6, 3), (64, 128, 256, 512)
EMBED, OUT, TOKENS, RES = 2048, 1024, 50, 224


def conv_names():
    """Convolution / BatchNorm key prefixes in the order ``lla_rn50_conv_desc`` enumerates them."""
    names = [("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3")]
    for s, nb in enumerate(BLOCKS):
        for b in range(nb):
            p = f"layer{s + 1}.{b}."
            names += [(p + "conv1", p + "bn1"), (p + "conv2", p + "bn2"), (p + "conv3", p + "bn3")]
            if b == 0:
                names.append((p + "downsample.1", p + "downsample.2"))
    return names


def rn50_macs_per_image():
    """Exact multiply-accumulate count of one 224x224 image through CLIP's ModifiedResNet-50 + AttentionPool2d
    (clip/model.py as loaded at lossyless/architectures.py:367-371): every convolution at the resolution it runs
    at (stem at 112x112, average pools before the strided blocks' conv3 / downsample), the attention pool's
    projections (q on the mean token, k and v on all 50) and its 32-head single-query attention, c_proj.
    -> (total MACs, {stage: MACs}).  Unpadded: the algorithmic work, the numerator of the roofline."""
    stages = {}
    stages["stem"] = 112 * 112 * (32 * 3 * 9 + 32 * 32 * 9 + 64 * 32 * 9)
    res, inplanes = 56, 64
    for s, nb in enumerate(BLOCKS):
        p, macs = PLANES[s], 0
        for b in range(nb):
            stride = 2 if (s > 0 and b == 0) else 1
            out = res // stride
            macs += res * res * (p * inplanes + p * p * 9)     # conv1 (1x1) and conv2 (3x3) at the input resolution
            macs += out * out * (4 * p * p)                    # conv3 (1x1) behind the average pool
            if b == 0:
                macs += out * out * (4 * p * inplanes)         # downsample: average pool + 1x1
            inplanes, res = 4 * p, out
        stages[f"layer{s + 1}"] = macs
    stages["attnpool"] = EMBED * EMBED + TOKENS * 2 * EMBED * EMBED + 2 * TOKENS * EMBED + OUT * EMBED
    return sum(stages.values()), stages


def synthetic_rn50_state_dict(seed=1):
    """Random-init RN50-CLIP visual weights in the OpenAI layout: He-normal convolutions, BatchNorm
    gamma ~ 1 (0.5 on the last BN of a block so that the residual stream stays O(1)), beta / running
    mean small, running var ~ 1.  No real checkpoint can be fetched offline."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5

    def bn(name, c, gamma=1.0):
        sd[name + ".weight"] = gamma * (1 + 0.1 * torch.randn(c, generator=g))
        sd[name + ".bias"] = 0.05 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.05 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1 + 0.1 * torch.rand(c, generator=g)

    conv("conv1", 32, 3, 3)
    bn("bn1", 32)
    conv("conv2", 32, 32, 3)
    bn("bn2", 32)
    conv("conv3", 64, 32, 3)
    bn("bn3", 64)
    inplanes = 64
    for s, nb in enumerate(BLOCKS):
        p = PLANES[s]
        for b in range(nb):
            pre = f"layer{s + 1}.{b}."
            conv(pre + "conv1", p, inplanes, 1)
            bn(pre + "bn1", p)
            conv(pre + "conv2", p, p, 3)
            bn(pre + "bn2", p)
            conv(pre + "conv3", 4 * p, p, 1)
            bn(pre + "bn3", 4 * p, gamma=0.5)
            if b == 0:
                conv(pre + "downsample.1", 4 * p, inplanes, 1)
                bn(pre + "downsample.2", 4 * p)
            inplanes = 4 * p
    sd["attnpool.positional_embedding"] = torch.randn(TOKENS, EMBED, generator=g) / EMBED ** 0.5
    for n, o in (("q_proj", EMBED), ("k_proj", EMBED), ("v_proj", EMBED), ("c_proj", OUT)):
        sd[f"attnpool.{n}.weight"] = torch.randn(o, EMBED, generator=g) / EMBED ** 0.5
        sd[f"attnpool.{n}.bias"] = 0.02 * torch.randn(o, generator=g)
    return sd


def fold_bn(sd, conv, bn, eps=1e-5):
    """-> (weight [cout, cin, k, k] fp32, bias [cout] fp32) of conv followed by eval-mode BatchNorm."""
    w = sd[conv + ".weight"].float()
    scale = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + eps)
    return w * scale.view(-1, 1, 1, 1), sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * scale


def pack_weights(sd):
    """OpenAI-layout state-dict -> uint8 numpy blob in the library's layout."""
    L = _lib.lib()
    blob = np.zeros(int(L.lla_rn50_weights_bytes()), dtype=np.uint8)
    names = conv_names()
    assert len(names) == int(L.lla_rn50_conv_count())
    d = (ctypes.c_int64 * 8)()
    for i, (cv, bn) in enumerate(names):
        _lib.check(L.lla_rn50_conv_desc(i, d), "lla_rn50_conv_desc")
        cin, cout, k, _, kpad, npad, w_off, b_off = (int(v) for v in d)
        w, b = fold_bn(sd, cv, bn)
        assert tuple(w.shape) == (cout, cin, k, k), (cv, tuple(w.shape), (cout, cin, k, k))
        wk = torch.zeros(npad, kpad)
        wk[:cout, : cin * k * k] = w.permute(0, 2, 3, 1).reshape(cout, -1)      # K order (kh, kw, c)
        bb = torch.zeros(npad)
        bb[:cout] = b
        blob[w_off:w_off + npad * kpad * 2] = wk.half().numpy().reshape(-1).view(np.uint8)
        blob[b_off:b_off + npad * 4] = bb.numpy().view(np.uint8)
    # first block of every stage: [W3 | Wds] and b3 + bds for the one-GEMM form of conv3 + downsample (lla_rn50_fused_desc)
    for s in range(len(BLOCKS)):
        _lib.check(L.lla_rn50_fused_desc(s, d), "lla_rn50_fused_desc")
        cin, cout, planes, inplanes, kpad, npad, w_off, b_off = (int(v) for v in d)
        pre = f"layer{s + 1}.0."
        w3, b3 = fold_bn(sd, pre + "conv3", pre + "bn3")
        wd, bd = fold_bn(sd, pre + "downsample.1", pre + "downsample.2")
        assert tuple(w3.shape) == (cout, planes, 1, 1) and tuple(wd.shape) == (cout, inplanes, 1, 1) and cin == planes + inplanes
        wk = torch.zeros(npad, kpad)
        wk[:cout, :planes] = w3.reshape(cout, planes)
        wk[:cout, planes:cin] = wd.reshape(cout, inplanes)
        bb = torch.zeros(npad)
        bb[:cout] = b3 + bd
        blob[w_off:w_off + npad * kpad * 2] = wk.half().numpy().reshape(-1).view(np.uint8)
        blob[b_off:b_off + npad * 4] = bb.numpy().view(np.uint8)
    o = (ctypes.c_int64 * 7)()
    _lib.check(L.lla_rn50_attnpool_offsets(o), "lla_rn50_attnpool_offsets")
    pos, qw, qb, kvw, kvb, cw, cb = (int(v) for v in o)

    def put(off, t, half):
        t = t.detach().float().contiguous()
        raw = (t.half().numpy() if half else t.numpy()).reshape(-1).view(np.uint8)
        blob[off:off + raw.nbytes] = raw

    put(pos, sd["attnpool.positional_embedding"].half().float(), False)
    put(qw, sd["attnpool.q_proj.weight"], True)
    put(qb, sd["attnpool.q_proj.bias"].half().float(), False)
    put(kvw, torch.cat([sd["attnpool.k_proj.weight"], sd["attnpool.v_proj.weight"]]), True)
    put(kvb, torch.cat([sd["attnpool.k_proj.bias"], sd["attnpool.v_proj.bias"]]).half().float(), False)
    put(cw, sd["attnpool.c_proj.weight"], True)
    put(cb, sd["attnpool.c_proj.bias"].half().float(), False)
    return blob


class ModifiedResNet(nn.Module):
    """``model.visual`` replacement for CLIP RN50: ``forward(X) -> z [B, 1024]`` fp16 with X
    [B,3,224,224] (NCHW, what clip feeds; converted to NHWC on the device) or [B,224,224,3] fp16."""

    def __init__(self, state_dict, chunk=256):
        super().__init__()
        self.register_buffer("blob", torch.from_numpy(pack_weights(state_dict)), persistent=False)
        self.chunk = int(chunk)
        self._ws = None
        self._tower = None
        self.input_resolution = RES
        self.output_dim = OUT

    def __getstate__(self):   # (see VisionTransformer.__getstate__: the handle and workspace are per-process scratch)
        state = self.__dict__.copy()
        state["_ws"] = None
        state["_tower"] = None
        return state

    def _apply(self, fn, recurse=True):
        _lib.upload_in_pieces(self, "blob", fn)   # (see VisionTransformer._apply)
        return super()._apply(fn, recurse)

    def forward(self, X, out=None):
        if X.dim() != 4:
            raise ValueError("expected a 4-D image batch")
        if tuple(X.shape[1:]) == (3, RES, RES):
            X = X.permute(0, 2, 3, 1)
        elif tuple(X.shape[1:]) != (RES, RES, 3):
            raise ValueError(f"expected [B,3,{RES},{RES}] or [B,{RES},{RES},3], got {tuple(X.shape)}")
        X = X.half().contiguous()
        _lib.require_cuda(X, "X")
        L = _lib.lib()
        B = X.shape[0]
        chunk = max(1, min(self.chunk, B))
        need = int(L.lla_rn50_workspace_bytes(chunk))
        if self._ws is None or self._ws.device != X.device or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=X.device)
        z = out if out is not None else torch.empty((B, OUT), dtype=torch.float16, device=X.device)
        dev = X.device if X.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            if self._tower is None or self._tower.device != dev:
                self._tower = _lib.Tower(dev)
            rc = L.lla_rn50_forward(_lib.ptr(X), B, _lib.ptr(self.blob), _lib.ptr(self._ws), self._ws.numel(),
                                    chunk, _lib.ptr(z), _lib.stream_ptr(dev), self._tower.handle)
        _lib.check(rc, "lla_rn50_forward")
        return z
