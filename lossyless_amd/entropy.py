"""EntropyBottleneck for the compress_dataset hot path, backed by HIP kernels.

Host-side mirror of ``compressai.entropy_models.EntropyBottleneck`` (compressai==1.1.5)
restricted to what ``hub/compressor.py`` uses: ctor ``(channels, init_scale, filters)``
(hub/compressor.py:49-51), ``update()`` (:63), ``forward`` in eval mode (:100),
``compress`` (:98) and ``decompress(strings, [1, 1])`` (:124), with the same parameter /
buffer names so the reference's state-dicts load unchanged.

Where the reference loops over images in Python and calls the C++ coder once per image
(re-marshalling the whole CDF table each time), this class hands the whole batch to
``lla_quantise_encode`` / ``lla_rans_decode_batch``: one image per GPU lane.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


class _LowerBound(nn.Module):
    """Holds the ``bound`` buffer (state-dict key ``likelihood_lower_bound.bound``)."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.tensor([float(bound)]))

    def forward(self, x):
        return torch.max(x, self.bound.to(x.dtype))


class HipEntropyCoder:
    """What ``EntropyModel.entropy_coder`` holds in this build (compressai keeps an ``_EntropyCoder("ans")`` there:
    the pybind11 rANS encoder / decoder pair).  The C-ABI is stateless, so this is only the marker the reference's
    ``make_pickable_`` / ``undo_pickable_`` / ``is_coder_present`` protocol toggles (lossyless/rates.py:273-284,
    :492-500): ``None`` = "no coder attached", and ``compress`` / ``decompress`` then refuse, as compressai's
    models fail on a ``None`` coder."""
    name = "hip-rans"

    def __repr__(self):
        return "HipEntropyCoder()"


def _require_coder(model):
    if getattr(model, "entropy_coder", None) is None:
        raise RuntimeError("no entropy coder attached (make_pickable_() was called): call undo_pickable_() / "
                           "prepare_compressor_() first")


def pmf_to_quantized_cdf(pmf, precision=16):
    """``compressai._CXX.pmf_to_quantized_cdf`` -> ``lla_pmf_to_quantized_cdf`` (host C-ABI)."""
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(pmf.shape[0] + 1, dtype=np.uint32)
    rc = _lib.lib().lla_pmf_to_quantized_cdf(pmf.ctypes.data_as(ctypes.c_void_p), pmf.shape[0],
                                             precision, out.ctypes.data_as(ctypes.c_void_p))
    _lib.check(rc, "lla_pmf_to_quantized_cdf")
    return out


def update_registered_buffers(module, module_name, buffer_names, state_dict,
                              policy="resize_if_empty", dtype=torch.int):
    """``compressai.models.utils.update_registered_buffers`` (hub/compressor.py:56-61):
    resize the dynamically sized CDF buffers so that ``load_state_dict`` accepts them."""
    for name in buffer_names:
        key = f"{module_name}.{name}"
        if key not in state_dict:
            raise KeyError(f"{key} missing from the state dict")
        new_size = state_dict[key].size()
        current = getattr(module, name)
        if policy == "resize_if_empty" and current.numel() != 0:
            raise RuntimeError(f"buffer {name} was not empty")
        # compressai resizes the REGISTERED buffer (keeping its dtype; `dtype` only applies to policy="register"):
        # `scale_table` of a GaussianConditional stays float -- re-registering it as int truncated the 64 scale
        # levels on load (found in round 4 by the fp32-index test: a loaded module built other indexes)
        # ... on the device the module already lives on (a module moved to the GPU before load_state_dict keeps
        # its tables there; `decompress()` takes its device from them)
        module.register_buffer(name, torch.zeros(new_size, dtype=current.dtype, device=current.device))


class EntropyBottleneck(nn.Module):
    entropy_coder_precision = 16

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3),
                 likelihood_bound=1e-9):
        super().__init__()
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)

        dims = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / dims[i + 1]))
            self.register_parameter(
                f"_matrix{i}", nn.Parameter(torch.full((channels, dims[i + 1], dims[i]), init)))
            self.register_parameter(
                f"_bias{i}", nn.Parameter(torch.empty(channels, dims[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(
                    f"_factor{i}", nn.Parameter(torch.zeros(channels, dims[i + 1], 1)))
        self.quantiles = nn.Parameter(
            torch.tensor([-self.init_scale, 0.0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.tensor([-target, 0.0, target], dtype=torch.float32))
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self.likelihood_lower_bound = _LowerBound(likelihood_bound)
        self.entropy_coder = HipEntropyCoder()
        self._dev_cache = None

    # ------------------------------------------------------------------ model
    def _logits_cumulative(self, inputs, params=None):
        """logits = stack of (softplus(matrix) @ x + bias, tanh gate); ``params`` lets
        ``update()`` evaluate it on CPU copies."""
        get = (lambda n: getattr(self, n).detach()) if params is None else (lambda n: params[n])
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(F.softplus(get(f"_matrix{i}")), logits)
            logits = logits + get(f"_bias{i}")
            if i < len(self.filters):
                logits = logits + torch.tanh(get(f"_factor{i}")) * torch.tanh(logits)
        return logits

    def _likelihood(self, inputs):
        lower = self._logits_cumulative(inputs - 0.5)
        upper = self._logits_cumulative(inputs + 0.5)
        sign = -torch.sign(lower + upper).detach()
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))

    def _medians(self):
        return self.quantiles[:, 0, 1].detach()

    def update(self, force=False):
        """Build the integer coding tables from the fp32 parameters.

        Like the reference this runs in fp32 on the CPU (there, because ``update()`` is
        called before ``.to(device)``, hub/compressor.py:63-66) and is skipped when the
        state-dict already carried tables -- which is how the shipped assets freeze them
        (SURVEY.md F5/F6)."""
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        cpu = torch.device("cpu")
        q = self.quantiles.detach().to(cpu, torch.float32)
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]

        params = {n: p.detach().to(cpu, torch.float32) for n, p in self.named_parameters()}
        logits = lambda t: self._logits_cumulative(t, params)
        lower = logits(samples - 0.5)
        upper = logits(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])

        cdf = torch.zeros((self.channels, max_length + 2), dtype=torch.int32)
        for c in range(self.channels):
            n = int(pmf_length[c])
            prob = torch.cat((pmf[c, :n], tail[c]), dim=0).numpy()
            row = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[c, : n + 2] = torch.from_numpy(row.astype(np.int32))
        self._quantized_cdf = cdf.to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._offset = (-minima).int().to(dev)
        self._dev_cache = None
        return True

    # ------------------------------------------------------------ device side
    def _apply(self, fn, *a, **k):
        self._dev_cache = None
        return super()._apply(fn, *a, **k)

    def device_tables(self, scaling=None, biasing=None):
        """Contiguous device copies of everything the kernels read.  ``scaling`` /
        ``biasing`` are the compressor's per-dimension affine (hub/compressor.py:46-47);
        exp(scaling) is evaluated once in float64 and rounded to fp32 so that it does not
        depend on which libm / vector width happens to run it."""
        if self._offset.numel() == 0:
            raise RuntimeError("EntropyBottleneck.update() has not been called")
        dev = self._quantized_cdf.device
        key = (dev, None if scaling is None else scaling._version,
               None if biasing is None else biasing._version, self.quantiles._version)
        if self._dev_cache is not None and self._dev_cache[0] == key:
            return self._dev_cache[1]
        C = self.channels
        es = (torch.ones(C, dtype=torch.float64) if scaling is None
              else torch.exp(scaling.detach().to("cpu", torch.float64))).to(torch.float32)
        bias = torch.zeros(C) if biasing is None else biasing.detach().to("cpu", torch.float32)
        t = dict(
            cdf=self._quantized_cdf.to(torch.int32).contiguous(),
            cdf_len=self._cdf_length.to(torch.int32).contiguous(),
            offset=self._offset.to(torch.int32).contiguous(),
            median=self._medians().to(dev, torch.float32).contiguous(),
            exp_scale=es.to(dev).contiguous(),
            bias=bias.to(dev).contiguous(),
            W=int(self._quantized_cdf.shape[1]),
        )
        self._dev_cache = (key, t)
        return t

    @staticmethod
    def _as_matrix(x):
        """[B, C, 1, 1] (what hub/compressor.py:109 builds) or [B, C] -> contiguous [B, C]."""
        if x.dim() == 4:
            if x.shape[2] != 1 or x.shape[3] != 1:
                raise ValueError("only 1x1 spatial latents are on this path (hub/compressor.py:124)")
            x = x.reshape(x.shape[0], x.shape[1])
        elif x.dim() != 2:
            raise ValueError("expected [B, C, 1, 1] or [B, C]")
        return x.contiguous()

    def encode_device(self, z, tables, want_symbols=False, record_prefix=False):
        """z [B, C] fp16/fp32 on the GPU -> (payload uint8 tensor, offsets uint64-as-int64
        tensor [B+1], symbols or None), all on the GPU, nothing synchronised."""
        _lib.require_cuda(z, "z")
        L = _lib.lib()
        B, C = z.shape
        dev = z.device
        if z.dtype == torch.float16:
            zt = _lib.LLA_Z_F16
        elif z.dtype == torch.float32:
            zt = _lib.LLA_Z_F32
        else:
            raise TypeError("z must be float16 or float32")
        stride = int(L.lla_rans_max_encoded_bytes(C))
        scratch = torch.empty(max(B, 1) * stride, dtype=torch.uint8, device=dev)
        lengths = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
        symbols = torch.empty((B, C), dtype=torch.int32, device=dev) if want_symbols else None
        st = _lib.stream_ptr(dev)
        rc = L.lla_quantise_encode(_lib.ptr(z), zt, B, C, _lib.ptr(tables["bias"]),
                                   _lib.ptr(tables["exp_scale"]), _lib.ptr(tables["median"]),
                                   _lib.ptr(tables["cdf"]), tables["W"], _lib.ptr(tables["cdf_len"]),
                                   _lib.ptr(tables["offset"]), _lib.ptr(scratch), stride,
                                   _lib.ptr(lengths), _lib.ptr(symbols), st)
        _lib.check(rc, "lla_quantise_encode")
        payload, offsets = self.compact_device(scratch, stride, lengths, B, record_prefix)
        return payload, offsets, symbols

    @staticmethod
    def compact_device(scratch, stride, lengths, B, record_prefix=False, cap=None):
        L = _lib.lib()
        dev = scratch.device
        if cap is None:
            cap = max(B, 1) * (stride + 4)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        offsets = torch.empty(B + 1, dtype=torch.int64, device=dev)
        wsb = int(L.lla_rans_compact_workspace_bytes(B))
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = L.lla_rans_compact(_lib.ptr(scratch), stride, _lib.ptr(lengths), B,
                                1 if record_prefix else 0, _lib.ptr(out), cap, _lib.ptr(offsets),
                                _lib.ptr(ws), wsb, _lib.stream_ptr(dev))
        _lib.check(rc, "lla_rans_compact")
        return out, offsets

    def decode_device(self, payload, offsets, B, tables, record_prefix=False):
        """payload uint8 / offsets int64 [B+1] on the GPU -> (symbols int32 [B, C], status)."""
        L = _lib.lib()
        C = self.channels
        dev = payload.device
        sym = torch.empty((B, C), dtype=torch.int32, device=dev)
        status = torch.zeros(max(B, 1), dtype=torch.int32, device=dev)
        rc = L.lla_rans_decode_batch(_lib.ptr(payload), _lib.ptr(offsets), 1 if record_prefix else 0,
                                     B, C, _lib.ptr(tables["cdf"]), tables["W"],
                                     _lib.ptr(tables["cdf_len"]), _lib.ptr(tables["offset"]),
                                     _lib.ptr(sym), _lib.ptr(status), _lib.stream_ptr(dev))
        _lib.check(rc, "lla_rans_decode_batch")
        return sym, status

    # ------------------------------------------------- compressai-shaped API
    def forward(self, x):
        """Eval mode: ``(round(x - median) + median, likelihood)``."""
        if self.training:
            raise NotImplementedError("training-mode noise is outside the compress_dataset path")
        perm = list(range(x.dim()))
        perm[0], perm[1] = 1, 0
        xp = x.permute(*perm)                       # [C, B, ...]
        values = xp.reshape(self.channels, 1, -1)
        med = self._medians().to(values.dtype)[:, None, None]
        outputs = torch.round(values - med) + med
        lik = self.likelihood_lower_bound(self._likelihood(outputs))
        return outputs.reshape(xp.shape).permute(*perm), lik.reshape(xp.shape).permute(*perm)

    def compress(self, x):
        """x [B, C, 1, 1] fp32 on the GPU -> list of B ``bytes`` (EntropyModel.compress)."""
        _require_coder(self)
        z = self._as_matrix(x)
        if z.dtype not in (torch.float16, torch.float32):
            z = z.float()
        tables = self.device_tables()  # median only: the affine was applied by the caller
        payload, offsets, _ = self.encode_device(z, tables)
        off = offsets.cpu().numpy()
        blob = payload[: int(off[-1])].cpu().numpy().tobytes()
        return [blob[int(off[i]):int(off[i + 1])] for i in range(z.shape[0])]

    def decompress(self, strings, size=(1, 1)):
        """list of ``bytes`` -> [B, C, 1, 1] fp32 (EntropyModel.decompress + dequantize)."""
        _require_coder(self)
        if tuple(size) != (1, 1):
            raise ValueError("only 1x1 spatial latents are on this path")
        dev = self._quantized_cdf.device
        if dev.type != "cuda":
            raise RuntimeError("decompress runs on the GPU only (no CPU fallback)")
        B = len(strings)
        tables = self.device_tables()
        lens = np.fromiter((len(s) for s in strings), dtype=np.int64, count=B)
        off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        blob = np.frombuffer(b"".join(strings) + b"\0\0\0\0", dtype=np.uint8).copy()
        payload = torch.from_numpy(blob).to(dev)
        offsets = torch.from_numpy(off).to(dev)
        sym, status = self.decode_device(payload, offsets, B, tables)
        if B and int(status.max()) != 0:
            raise ValueError("malformed rANS stream")
        out = sym.to(torch.float32) + tables["median"][None, :]
        return out.reshape(B, self.channels, 1, 1)


class GaussianConditional(nn.Module):
    """Host-side mirror of ``compressai.entropy_models.GaussianConditional`` (compressai==1.1.5)
    for the hyperprior coder of the reference (``lossyless/rates.py:572-729``: ``HRateHyperprior``
    holds ``GaussianConditional(None)``, sets the 64-level table of ``get_scale_table`` through
    ``update_scale_table`` (:296-299), picks table rows with ``build_indexes`` (:698) and codes with
    ``compress(z, indexes, means=)`` / ``decompress`` (:712, :722)) -- SURVEY.md 8(f) rank 4.

    Same buffer names (``scale_table``, ``_quantized_cdf``, ``_offset``, ``_cdf_length``,
    ``scale_bound``).  Coding goes through ``lla_rans_encode_indexed`` /
    ``lla_rans_decode_indexed``: one string per GPU lane, table rows read from HBM/L2."""

    entropy_coder_precision = 16

    def __init__(self, scale_table, scale_bound=0.11, tail_mass=1e-9, likelihood_bound=1e-9):
        super().__init__()
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        if scale_table and (list(scale_table) != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = float(scale_table[0])
        if scale_bound is None or scale_bound <= 0:
            raise ValueError("Invalid parameters")
        self.lower_bound_scale = _LowerBound(scale_bound)
        self.likelihood_lower_bound = _LowerBound(likelihood_bound)
        self.register_buffer("scale_table", self._prepare_scale_table(scale_table) if scale_table
                             else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self.entropy_coder = HipEntropyCoder()
        self._dev_tables = None

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_cumulative(inputs):
        # the complementary error function maximises precision in the tails
        return 0.5 * torch.erfc(-(2 ** -0.5) * inputs)

    @staticmethod
    def _standardized_quantile(quantile):
        import scipy.stats
        return scipy.stats.norm.ppf(quantile)

    def update_scale_table(self, scale_table, force=False):
        """True if the tables were (re)built; they are kept when already present unless ``force``."""
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = self._prepare_scale_table(scale_table).to(device)
        self.update()
        return True

    def update(self):
        """fp32 on the CPU, like ``EntropyBottleneck.update``; rows go through the host C-ABI
        ``lla_pmf_to_quantized_cdf``."""
        device = self.scale_table.device
        scale_table = self.scale_table.detach().float().cpu()
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = scale_table.unsqueeze(1)
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i in range(len(pmf_length)):
            n = int(pmf_length[i])
            prob = torch.cat((pmf[i, :n], tail_mass[i]), dim=0)
            row = pmf_to_quantized_cdf(prob.numpy(), self.entropy_coder_precision)
            cdf[i, :row.shape[0]] = torch.from_numpy(row.astype(np.int64)).to(torch.int32)
        self._quantized_cdf = cdf.to(device)
        self._offset = (-pmf_center).to(device)
        self._cdf_length = (pmf_length + 2).to(device)
        self._dev_tables = None

    def _apply(self, fn, *a, **k):
        self._dev_tables = None
        return super()._apply(fn, *a, **k)

    def device_tables(self):
        if self._offset.numel() == 0:
            raise RuntimeError("call update_scale_table() first")
        if self._dev_tables is None:
            self._dev_tables = dict(cdf=self._quantized_cdf.to(torch.int32).contiguous(),
                                    cdf_len=self._cdf_length.to(torch.int32).contiguous(),
                                    offset=self._offset.to(torch.int32).contiguous(),
                                    T=int(self._quantized_cdf.shape[0]), W=int(self._quantized_cdf.shape[1]))
        return self._dev_tables

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes

    def _likelihood(self, inputs, scales, means=None):
        values = inputs - means if means is not None else inputs
        scales = self.lower_bound_scale(scales)
        values = torch.abs(values)
        upper = self._standardized_cumulative((0.5 - values) / scales)
        lower = self._standardized_cumulative((-0.5 - values) / scales)
        return upper - lower

    def forward(self, inputs, scales, means=None):
        """Eval mode: ``(round(inputs - means) + means, likelihood)``."""
        if self.training:
            raise NotImplementedError("training-mode noise is outside the coding path")
        outputs = torch.round(inputs - means) + means if means is not None else torch.round(inputs)
        likelihood = self.likelihood_lower_bound(self._likelihood(outputs, scales, means))
        return outputs, likelihood

    @staticmethod
    def _rows(t):
        return t.reshape(t.shape[0], -1).contiguous()

    @torch.no_grad()
    def compress(self, inputs, indexes, means=None):
        """inputs / indexes / means [B, ...] on the GPU -> list of B ``bytes``
        (``EntropyModel.compress``: symbols = round(inputs - means), one string per row)."""
        _require_coder(self)
        _lib.require_cuda(inputs, "inputs")
        tables = self.device_tables()
        vals = inputs.float() - means.float() if means is not None else inputs.float()
        sym = self._rows(torch.round(vals).to(torch.int32))
        idx = self._rows(indexes.to(torch.int32))
        if idx.shape != sym.shape:
            raise ValueError("`inputs` and `indexes` should have the same size.")
        B, n = sym.shape
        L = _lib.lib()
        dev = sym.device
        stride = int(L.lla_rans_max_encoded_bytes(n))
        scratch = torch.empty(max(B, 1) * stride, dtype=torch.uint8, device=dev)
        lengths = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
        rc = L.lla_rans_encode_indexed(_lib.ptr(sym), _lib.ptr(idx), B, n, _lib.ptr(tables["cdf"]),
                                       tables["T"], tables["W"], _lib.ptr(tables["cdf_len"]),
                                       _lib.ptr(tables["offset"]), _lib.ptr(scratch), stride,
                                       _lib.ptr(lengths), _lib.stream_ptr(dev))
        _lib.check(rc, "lla_rans_encode_indexed")
        payload, offsets = EntropyBottleneck.compact_device(scratch, stride, lengths, B)
        off = offsets.cpu().numpy()
        blob = payload[: int(off[-1])].cpu().numpy().tobytes()
        return [blob[int(off[i]):int(off[i + 1])] for i in range(B)]

    @torch.no_grad()
    def decompress(self, strings, indexes, means=None):
        """list of B ``bytes`` + indexes [B, ...] (+ means) -> fp32 tensor shaped like ``indexes``."""
        _require_coder(self)
        _lib.require_cuda(indexes, "indexes")
        tables = self.device_tables()
        idx = self._rows(indexes.to(torch.int32))
        B, n = idx.shape
        if len(strings) != B:
            raise ValueError("one string per row of `indexes` expected")
        dev = idx.device
        lens = np.fromiter((len(s) for s in strings), dtype=np.int64, count=B)
        off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        blob = np.frombuffer(b"".join(strings) + b"\0\0\0\0", dtype=np.uint8).copy()
        sym = torch.empty((B, n), dtype=torch.int32, device=dev)
        status = torch.zeros(max(B, 1), dtype=torch.int32, device=dev)
        payload = torch.from_numpy(blob).to(dev)   # named: both must outlive the launch
        offsets = torch.from_numpy(off).to(dev)
        rc = _lib.lib().lla_rans_decode_indexed(
            _lib.ptr(payload), _lib.ptr(offsets), 0, B, n,
            _lib.ptr(idx), _lib.ptr(tables["cdf"]), tables["T"], tables["W"],
            _lib.ptr(tables["cdf_len"]), _lib.ptr(tables["offset"]), _lib.ptr(sym), _lib.ptr(status),
            _lib.stream_ptr(dev))
        _lib.check(rc, "lla_rans_decode_indexed")
        if B and int(status.max()) != 0:
            raise ValueError("malformed rANS stream")
        out = sym.to(torch.float32).reshape(indexes.shape)
        return out + means.float() if means is not None else out
