"""Training-side twin of the hub compressor's coder: ``HRateFactorizedPrior``.

Mirror of the subset of ``lossyless/rates.py`` that shares the hub path's arithmetic
(``HRateEstimator`` :398-506 -- ``scaling/biasing``, ``process_z_in/out`` :434-438, the
CDF-buffer-aware load hook :440-473 -- and ``HRateFactorizedPrior`` :509-564, plus
``RateEstimator.real_rate / update / prepare_compressor_`` :215-314), so that the reference's
evaluation code (``learnable_compressors.py:339-341``) can code representations with the HIP
kernels.  Only inference-time coding is provided; the training losses stay in the reference.
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .entropy import EntropyBottleneck, update_registered_buffers


class HRateFactorizedPrior(nn.Module):
    """z [B, z_dim] -> one rANS stream per row, same bytes as ``ClipCompressor.compress``.

    ``compress`` returns ``[strings]`` (a list holding the list of byte strings: the
    reference keeps the outer list "for generality when hyperprior", rates.py:556-559)."""

    is_can_compress = True

    def __init__(self, z_dim, kwargs_ent_bottleneck={}, **kwargs):
        super().__init__()
        self.z_dim = z_dim
        self.kwargs_ent_bottleneck = dict(kwargs_ent_bottleneck)
        self.scaling = torch.nn.Parameter(torch.ones(z_dim))
        self.biasing = torch.nn.Parameter(torch.zeros(z_dim))
        self.entropy_bottleneck = EntropyBottleneck(z_dim, **self.kwargs_ent_bottleneck)

    # rates.py:434-438
    def process_z_in(self, z):
        return (z.float() + self.biasing) * self.scaling.exp()

    def process_z_out(self, z_hat):
        return (z_hat / self.scaling.exp()) - self.biasing

    # rates.py:440-473: the CDF buffers have data-dependent sizes
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        try:
            update_registered_buffers(self.entropy_bottleneck, f"{prefix}entropy_bottleneck",
                                      ["_quantized_cdf", "_offset", "_cdf_length"], state_dict,
                                      policy="resize")
        except KeyError:
            pass
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @property
    def is_coder_updated(self):
        return self.entropy_bottleneck._offset.numel() > 0

    def update(self, force=False):
        """rates.py:286-305: (re)build the integer tables; True if they changed."""
        return self.entropy_bottleneck.update(force=force)

    def prepare_compressor_(self):
        """rates.py:307-314."""
        self.update(force=True)

    def _tables(self):
        return self.entropy_bottleneck.device_tables(self.scaling, self.biasing)

    @torch.no_grad()
    def compress(self, z, parent=None):
        """rates.py:556-559.  z [B, z_dim] on the GPU (fp16 or fp32)."""
        if not self.is_coder_updated:
            raise RuntimeError("call update() / prepare_compressor_() first")
        z = z.contiguous()
        if z.dtype not in (torch.float16, torch.float32):
            z = z.float()
        payload, offsets, _ = self.entropy_bottleneck.encode_device(z, self._tables())
        off = offsets.cpu().numpy()
        blob = payload[: int(off[-1])].cpu().numpy().tobytes()
        return [[blob[int(off[i]):int(off[i + 1])] for i in range(z.shape[0])]]

    @torch.no_grad()
    def decompress(self, all_strings):
        """rates.py:561-564 -> z_hat [B, z_dim] fp32 on the GPU."""
        assert isinstance(all_strings, list) and len(all_strings) == 1
        strings = all_strings[0]
        import numpy as np
        B = len(strings)
        lens = np.fromiter((len(s) for s in strings), dtype=np.int64, count=B)
        off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        dev = self.scaling.device
        blob = np.frombuffer(b"".join(strings) + b"\0\0\0\0", dtype=np.uint8).copy()
        tables = self._tables()
        sym, status = self.entropy_bottleneck.decode_device(
            torch.from_numpy(blob).to(dev), torch.from_numpy(off).to(dev), B, tables)
        if B and int(status.max()) != 0:
            raise ValueError("malformed rANS stream")
        out = torch.empty((B, self.z_dim), dtype=torch.float32, device=dev)
        rc = _lib.lib().lla_dequantise(_lib.ptr(sym), B, self.z_dim, _lib.ptr(tables["bias"]),
                                       _lib.ptr(tables["exp_scale"]), _lib.ptr(tables["median"]),
                                       _lib.ptr(out), _lib.stream_ptr(dev))
        _lib.check(rc, "lla_dequantise")
        return out

    def real_rate(self, z, is_return_logs=False, parent=None):
        """rates.py:215-260: mean coded bits per example (sum over latents, mean over batch);
        with ``is_return_logs`` also the reference's log dict -- ``compress_time`` and
        ``receiver_time`` in seconds per example (rates.py:241-257), ``n_bits``."""
        import time
        batch = z.shape[0]
        dev = z.device if z.is_cuda else None

        def clock():
            if dev is not None:
                torch.cuda.synchronize(dev)
            return time.perf_counter()

        t0 = clock()
        all_strings = self.compress(z, parent=parent)
        t1 = clock()
        if is_return_logs:
            _ = self.decompress(all_strings)
            t2 = clock()
        n_bytes = sum(sum(len(s) for s in strings) / len(strings) for strings in all_strings)
        n_bits = n_bytes * 8
        if is_return_logs:
            logs = dict(compress_time=(t1 - t0) / batch, receiver_time=(t2 - t1) / batch,
                        n_bits=n_bits)
            return n_bits, logs
        return n_bits


def get_scale_table(min=0.11, max=256, levels=64):
    """rates.py:567-569."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class MLP(nn.Module):
    """The reference's ``MLP`` (lossyless/architectures.py:94-168) in the configuration the
    hyperprior uses it (identity norm, ReLU, no dropout): same ``module`` Sequential layout, so
    its state-dict keys (``module.0.weight`` ... ) load unchanged.

    On the GPU the layers run on the library's MFMA GEMMs (``lla_gemm_f16_ex``: fp16 operands, fp32
    accumulation, bias + ReLU in the epilogue; dimensions zero-padded to the kernels' 128 / 64 granules:
    side_z_dim = 102 -> 128), not on torch / hipBLASLt.  The side information and the scale indexes are
    whatever THIS network computes, on the encoder and on the decoder alike (as with the reference, whose
    strings are only decodable by the arithmetic that wrote them); CPU tensors take torch's fp32 Linear."""

    def __init__(self, in_dim, out_dim, n_hid_layers=1, hid_dim=128):
        super().__init__()
        layers = [nn.Linear(in_dim, hid_dim), nn.Identity(), nn.ReLU(), nn.Identity()]
        for _ in range(1, n_hid_layers):
            layers += [nn.Linear(hid_dim, hid_dim), nn.Identity(), nn.ReLU(), nn.Identity()]
        layers += [nn.Linear(hid_dim, out_dim)]
        self.module = nn.Sequential(*layers)
        self._packed = None

    def _pack(self, dev):
        """fp16 [Npad][Kpad] weights + fp32 [Npad] biases per Linear, cached per device / parameter version."""
        lin = [m for m in self.module if isinstance(m, nn.Linear)]
        key = (str(dev),) + tuple((m.weight._version, m.bias._version, m.weight.data_ptr()) for m in lin)
        if self._packed is None or self._packed[0] != key:
            packs = []
            for m in lin:
                n, k = m.weight.shape
                npad, kpad = -(-n // 128) * 128, -(-k // 64) * 64
                w = torch.zeros((npad, kpad), dtype=torch.float16, device=dev)
                w[:n, :k] = m.weight.detach().to(dev, torch.float16)
                b = torch.zeros(npad, dtype=torch.float32, device=dev)
                b[:n] = m.bias.detach().to(dev, torch.float32)
                packs.append((w, b, n, k, npad, kpad))
            self._packed = (key, packs)
        return self._packed[1]

    def forward(self, X):
        shape = X.shape
        X2 = X.reshape(-1, shape[-1])
        if not X2.is_cuda:
            return self.module(X2).reshape(*shape[:-1], -1)
        from . import _lib
        L, dev = _lib.lib(), X2.device
        packs = self._pack(dev)
        rows = X2.shape[0]
        a = torch.zeros((rows, packs[0][5]), dtype=torch.float16, device=dev)
        a[:, :packs[0][3]] = X2.to(torch.float16)
        for i, (w, b, n, k, npad, kpad) in enumerate(packs):
            last = i == len(packs) - 1
            c = torch.empty((rows, npad), dtype=torch.float16, device=dev)
            rc = L.lla_gemm_f16_ex(_lib.ptr(a), a.shape[1], _lib.ptr(w), _lib.ptr(b), _lib.ptr(c), npad, None, 0,
                                   rows, npad, kpad, _lib.LLA_EPI_F16 if last else _lib.LLA_EPI_RELU_F16,
                                   _lib.stream_ptr(dev))
            _lib.check(rc, "lla_gemm_f16_ex")
            a = c          # (npad is a multiple of 128, hence a valid Kpad of the next layer; padding columns are 0)
        out_dim = packs[-1][2]
        return a[:, :out_dim].float().reshape(*shape[:-1], out_dim)


class HRateHyperprior(HRateFactorizedPrior):
    """Scale-hyperprior coder twin (``lossyless/rates.py:572-756``), inference-time coding only.

    ``compress(z)`` -> ``[z_strings, side_z_strings]``: the side information ``side_encoder(z_in)``
    goes through the factorized ``EntropyBottleneck`` (HIP, one row per lane, table row = channel),
    its decoded value drives ``z_encoder`` -> (scales, means) -> ``build_indexes``, and ``z_in`` is
    coded by ``GaussianConditional`` with those per-element table rows (HIP,
    ``lla_rans_encode_indexed``).  ``get_indexes_means_hat`` reproduces the reference verbatim,
    including that it passes the *scales* as ``means`` (rates.py:694-696 overwrite ``means_hat``
    with ``atleast_ndim(scales_hat, 4)``): bitstreams must match what the reference would write.
    """

    def __init__(self, z_dim, factor_dim=5, side_z_dim=None, is_pred_mean=True,
                 kwargs_ent_bottleneck={}, **kwargs):
        from .entropy import GaussianConditional
        super().__init__(z_dim, kwargs_ent_bottleneck=kwargs_ent_bottleneck, **kwargs)
        if side_z_dim is None:
            side_z_dim = max(10, z_dim // factor_dim)
        self.side_z_dim = side_z_dim
        self.is_pred_mean = is_pred_mean
        self.entropy_bottleneck = EntropyBottleneck(side_z_dim, **self.kwargs_ent_bottleneck)
        self.gaussian_conditional = GaussianConditional(None)
        kwargs_mlp = dict(n_hid_layers=2, hid_dim=max(z_dim, 256))
        self.side_encoder = MLP(z_dim, side_z_dim, **kwargs_mlp)
        self.z_encoder = MLP(side_z_dim, z_dim * (2 if is_pred_mean else 1), **kwargs_mlp)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        try:
            update_registered_buffers(self.gaussian_conditional, f"{prefix}gaussian_conditional",
                                      ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"],
                                      state_dict, policy="resize")
        except KeyError:
            pass
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @property
    def is_coder_updated(self):
        return (self.entropy_bottleneck._offset.numel() > 0
                and self.gaussian_conditional._offset.numel() > 0)

    def update(self, force=False):
        """rates.py:286-305."""
        updated = bool(self.entropy_bottleneck.update(force=force))
        updated &= bool(self.gaussian_conditional.update_scale_table(get_scale_table(), force=force))
        return updated

    def chunk_params(self, gaussian_params):
        if self.is_pred_mean:
            scales_hat, means_hat = gaussian_params.chunk(2, -1)
        else:
            scales_hat, means_hat = gaussian_params, None
        return scales_hat, means_hat

    def _side_strings_to_rows(self, strings):
        """EntropyBottleneck.decompress of the lossyless wrapper (rates.py:68-71): [B, side_z_dim]."""
        return self.entropy_bottleneck.decompress(strings).reshape(len(strings), self.side_z_dim)

    def get_indexes_means_hat(self, side_z_strings):
        side_z_hat = self._side_strings_to_rows(side_z_strings)
        gaussian_params = self.z_encoder(side_z_hat)
        scales_hat, means_hat = self.chunk_params(gaussian_params)
        scales_hat = scales_hat.reshape(*scales_hat.shape, 1, 1)
        means_hat = scales_hat  # (sic) rates.py:695-696
        indexes = self.gaussian_conditional.build_indexes(scales_hat)
        return indexes, means_hat

    @torch.no_grad()
    def compress(self, z, parent=None):
        """rates.py:701-713 -> ``[z_strings, side_z_strings]``."""
        if not self.is_coder_updated:
            raise RuntimeError("call update() / prepare_compressor_() first")
        z_in = self.process_z_in(z)
        side_z = self.side_encoder(z_in)
        side_z_strings = self.entropy_bottleneck.compress(side_z.contiguous())
        indexes, means_hat = self.get_indexes_means_hat(side_z_strings)
        z_in = z_in.reshape(*z_in.shape, 1, 1)
        z_strings = self.gaussian_conditional.compress(z_in, indexes, means=means_hat)
        return [z_strings, side_z_strings]

    @torch.no_grad()
    def decompress(self, all_strings):
        """rates.py:715-724 -> z_hat [B, z_dim] fp32 on the GPU."""
        assert isinstance(all_strings, list) and len(all_strings) == 2
        z_strings, side_z_strings = all_strings
        indexes, means_hat = self.get_indexes_means_hat(side_z_strings)
        z_hat = self.gaussian_conditional.decompress(z_strings, indexes, means=means_hat)
        return self.process_z_out(z_hat.reshape(z_hat.shape[0], -1))
