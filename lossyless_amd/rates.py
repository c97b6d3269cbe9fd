"""Training-side twin of the hub compressor's coder: ``HRateFactorizedPrior`` / ``HRateHyperprior``.

Mirror of the inference-time surface of ``lossyless/rates.py`` that the reference's evaluator drives
(``learnable_compressors.py:84`` ``make_pickable_``, ``:161-163,190`` ``compress`` / ``__call__``, ``:341``
``prepare_compressor_``, ``:436`` ``make_pickable_``): ``RateEstimator`` :77-314 (``forward`` ->
``forward_help`` -> ``(z_hat, rates, logs, other)``, ``real_rate``, ``make_pickable_`` / ``undo_pickable_``,
``update``, ``prepare_compressor_``), ``HRateEstimator`` :398-506 (``scaling/biasing``, ``process_z_in/out``
:434-438, the CDF-buffer-aware load hook :440-473, ``is_coder_updated / is_coder_present /
is_compute_real_rate`` :482-506), ``HRateFactorizedPrior`` :509-564 and ``HRateHyperprior`` :572-756 -- so that
``LearnableCompressor`` can evaluate and code representations with the HIP kernels.  Training-mode noise and the
auxiliary losses stay in the reference (out of scope).
"""
import math

import torch
import torch.nn as nn

from . import _lib
from .entropy import EntropyBottleneck, _require_coder, update_registered_buffers


BASE_LOG = 2   # lossyless/helpers.py:27


def _entropy_models(module):
    from .entropy import GaussianConditional
    return [m for m in module.modules() if isinstance(m, (EntropyBottleneck, GaussianConditional))]


class RateEstimator(nn.Module):
    """lossyless/rates.py:77-314 (base class of the coders), inference-time part."""

    is_can_compress = False

    def __init__(self, z_dim, warmup_k_epoch=0, is_endToEnd=True):
        super().__init__()
        self.z_dim = z_dim
        self.warmup_k_epoch = warmup_k_epoch
        self.is_endToEnd = is_endToEnd

    def forward(self, z, p_Zlx, parent=None):
        """rates.py:104-146 -> ``(z_hat, rates [batch], logs, other)``; fp32 whatever autocast says."""
        with torch.autocast(device_type=z.device.type, enabled=False):
            z_hat, rates, r_logs, r_other = self.forward_help(z, p_Zlx, parent)
            epoch = getattr(parent, "current_epoch", self.warmup_k_epoch)
            if (not self.is_endToEnd) or (epoch < self.warmup_k_epoch):
                # disjoint training / warm-up (rates.py:136-144): the rate is recomputed on detached inputs
                z_detached = z.detach() + z * 0
                p_detached = p_Zlx.detach(is_grad_flow=True) if p_Zlx is not None else None
                _, rates, *_ = self.forward_help(z_detached, p_detached, parent)
        return z_hat, rates, r_logs, r_other

    def forward_help(self, z, p_Zlx, parent=None):
        raise NotImplementedError()

    def compress(self, z, parent=None):
        raise NotImplementedError()

    def decompress(self, all_strings):
        raise NotImplementedError()

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

    def make_pickable_(self):
        """rates.py:273-277: detach the coder objects (the reference's are pybind11 handles that do not pickle;
        ``LearnableCompressor`` calls this right after construction and after the test epoch,
        learnable_compressors.py:84,436).  ``is_coder_present`` turns False, ``compress`` refuses."""
        for m in _entropy_models(self):
            m.entropy_coder = None

    def undo_pickable_(self):
        """rates.py:279-284."""
        from .entropy import HipEntropyCoder
        for m in _entropy_models(self):
            m.entropy_coder = HipEntropyCoder()

    def update(self, force=False):
        """rates.py:286-305: (re)build the integer tables of every entropy model; True if all changed."""
        from .entropy import GaussianConditional
        updated = True
        for m in self.children():
            if isinstance(m, EntropyBottleneck):
                updated &= bool(m.update(force=force))
            elif isinstance(m, GaussianConditional):
                updated &= bool(m.update_scale_table(get_scale_table(), force=force))
        return updated

    def prepare_compressor_(self):
        """rates.py:307-314: coder attached, tables rebuilt."""
        self.undo_pickable_()
        self.update(force=True)


class HRateEstimator(RateEstimator):
    """lossyless/rates.py:398-506: per-dimension affine in front of the entropy models + coder state."""

    is_can_compress = True

    def __init__(self, z_dim, kwargs_ent_bottleneck={}, **kwargs):
        super().__init__(z_dim, **kwargs)
        self.kwargs_ent_bottleneck = dict(kwargs_ent_bottleneck)
        self.scaling = torch.nn.Parameter(torch.ones(z_dim))
        self.biasing = torch.nn.Parameter(torch.zeros(z_dim))

    def _exp_scaling(self):
        """exp(scaling) as the coding tables hold it (``EntropyBottleneck.device_tables``): evaluated in float64,
        rounded once to fp32 -- so ``forward`` quantises exactly where ``compress`` does."""
        return torch.exp(self.scaling.double()).float()

    # rates.py:434-438
    def process_z_in(self, z):
        return (z.float() + self.biasing) * self._exp_scaling()

    def process_z_out(self, z_hat):
        return (z_hat / self._exp_scaling()) - self.biasing

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
        """rates.py:482-490."""
        return all(m._offset.numel() > 0 for m in _entropy_models(self))

    @property
    def is_coder_present(self):
        """rates.py:492-500."""
        return all(m.entropy_coder is not None for m in _entropy_models(self))

    @property
    def is_compute_real_rate(self):
        """rates.py:502-506."""
        return (not self.training) and self.is_coder_updated and self.is_coder_present

    def _add_real_rate(self, z, logs):
        """The tail both ``forward_help`` share (rates.py:542-545, :668-671)."""
        if self.is_compute_real_rate:
            n_bits, logs2 = self.real_rate(z, is_return_logs=True)
            logs.update(logs2)
            logs["n_bits"] = n_bits


class HRateFactorizedPrior(HRateEstimator):
    """z [B, z_dim] -> one rANS stream per row, same bytes as ``ClipCompressor.compress``.

    ``compress`` returns ``[strings]`` (a list holding the list of byte strings: the
    reference keeps the outer list "for generality when hyperprior", rates.py:556-559)."""

    def __init__(self, z_dim, **kwargs):
        super().__init__(z_dim, **kwargs)
        self.entropy_bottleneck = EntropyBottleneck(z_dim, **self.kwargs_ent_bottleneck)

    def forward_help(self, z, _, parent=None):
        """rates.py:534-554 (eval mode): ``z_hat`` = dequantised representation, ``-log q(z)`` per example in
        nats, logs ``H_q_Z`` (bits), ``H_ZlX`` and -- when the coder is usable -- the real rate's
        ``n_bits / compress_time / receiver_time``."""
        z_in = self.process_z_in(z)
        z_hat, q_z = self.entropy_bottleneck(z_in)
        neg_log_q_z = -torch.log(q_z).sum(-1)
        logs = dict(H_q_Z=neg_log_q_z.mean() / math.log(BASE_LOG), H_ZlX=0)
        self._add_real_rate(z, logs)
        other = dict()
        z_hat = self.process_z_out(z_hat)
        return z_hat, neg_log_q_z, logs, other

    def _tables(self):
        return self.entropy_bottleneck.device_tables(self.scaling, self.biasing)

    @torch.no_grad()
    def compress(self, z, parent=None):
        """rates.py:556-559.  z [B, z_dim] on the GPU (fp16 or fp32)."""
        if not self.is_coder_updated:
            raise RuntimeError("call update() / prepare_compressor_() first")
        _require_coder(self.entropy_bottleneck)
        z = z.contiguous()
        if z.dtype not in (torch.float16, torch.float32):
            z = z.float()
        if not z.is_cuda:
            return [self._compress_host(z)]
        payload, offsets, _ = self.entropy_bottleneck.encode_device(z, self._tables())
        off = offsets.cpu().numpy()
        blob = payload[: int(off[-1])].cpu().numpy().tobytes()
        return [[blob[int(off[i]):int(off[i + 1])] for i in range(z.shape[0])]]

    @torch.no_grad()
    def decompress(self, all_strings):
        """rates.py:561-564 -> z_hat [B, z_dim] fp32 on the GPU."""
        assert isinstance(all_strings, list) and len(all_strings) == 1
        _require_coder(self.entropy_bottleneck)
        strings = all_strings[0]
        import numpy as np
        if self.scaling.device.type != "cuda":
            return self._decompress_host(strings)
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


    # ---- a module on the CPU codes with the library's HOST coder (the reference's coder runs on the CPU
    # wherever the module lives): same symbols (three separately rounded fp32 operations, round-half-even),
    # same strings as the device kernels
    def _host_tables(self):
        import numpy as np
        t = self._tables()
        return {k: (np.ascontiguousarray(v.cpu().numpy()) if hasattr(v, "cpu") else v) for k, v in t.items()}

    def _compress_host(self, z):
        import ctypes
        import numpy as np
        t = self._host_tables()
        zf = z.float()
        y = (zf + torch.from_numpy(t["bias"])) * torch.from_numpy(t["exp_scale"])
        sym = np.ascontiguousarray(torch.round(y - torch.from_numpy(t["median"])).to(torch.int32).numpy())
        B, C = sym.shape
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        off = np.zeros(B + 1, dtype=np.uint64)
        L = _lib.lib()
        args = (P(sym), B, C, P(t["cdf"]), t["W"], P(t["cdf_len"]), P(t["offset"]), 0)
        rc = L.lla_rans_encode_batch_host(*args, None, 0, P(off))
        if rc not in (_lib.LLA_OK, -2):
            _lib.check(rc, "lla_rans_encode_batch_host")
        out = np.empty(max(int(off[-1]), 1), dtype=np.uint8)
        _lib.check(L.lla_rans_encode_batch_host(*args, P(out), out.size, P(off)), "lla_rans_encode_batch_host")
        blob = out.tobytes()
        return [blob[int(off[i]):int(off[i + 1])] for i in range(B)]

    def _decompress_host(self, strings):
        import ctypes
        import numpy as np
        t = self._host_tables()
        B, C = len(strings), self.z_dim
        lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=B)
        off = np.zeros(B + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        blob = np.frombuffer(b"".join(strings) + b"\0\0\0\0", dtype=np.uint8).copy()
        sym = np.empty((B, C), dtype=np.int32)
        status = np.zeros(max(B, 1), dtype=np.int32)
        out = np.empty((B, C), dtype=np.float32)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        L = _lib.lib()
        _lib.check(L.lla_rans_decode_batch_host(P(blob), P(off), 0, B, C, P(t["cdf"]), t["W"], P(t["cdf_len"]),
                                                P(t["offset"]), P(sym), P(status)), "lla_rans_decode_batch_host")
        if B and int(status.max()) != 0:
            raise ValueError("malformed rANS stream")
        _lib.check(L.lla_dequantise_host(P(sym), B, C, P(t["bias"]), P(t["exp_scale"]), P(t["median"]), P(out)),
                   "lla_dequantise_host")
        return torch.from_numpy(out)


def get_scale_table(min=0.11, max=256, levels=64):
    """rates.py:567-569."""
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class MLP(nn.Module):
    """The reference's ``MLP`` (lossyless/architectures.py:94-168) in the configuration the
    hyperprior uses it (identity norm, ReLU, no dropout): same ``module`` Sequential layout, so
    its state-dict keys (``module.0.weight`` ... ) load unchanged.

    On the GPU the layers run on the library's fp32 matrix-core GEMM (``lla_gemm_f32``:
    ``v_mfma_f32_32x32x2_f32``, fp32 operands and accumulation, bias + ReLU in the epilogue) -- the
    reference evaluates these networks in fp32 under ``autocast(False)`` (lossyless/rates.py:104: "precision here
    is important"), and the scale indexes they produce select the coding tables, i.e. they are part of the
    bitstream's contract.  The K order of every output element is fixed by the kernel, so the values do not depend
    on the batch size (a decoder may evaluate the network in other pieces than the encoder did: tested).  They are
    fp32-roundoff close to, not bit-identical with, a CPU / hipBLASLt evaluation (different summation order):
    strings written here are decoded here.  ``precision="fp16"`` (opt-in, round 3's path) runs the layers on the
    tower's fp16 MFMA GEMMs instead (``lla_gemm_f16_ex``); CPU tensors take torch's fp32 Linear."""

    def __init__(self, in_dim, out_dim, n_hid_layers=1, hid_dim=128, precision="fp32"):
        super().__init__()
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        self.precision = precision
        layers = [nn.Linear(in_dim, hid_dim), nn.Identity(), nn.ReLU(), nn.Identity()]
        for _ in range(1, n_hid_layers):
            layers += [nn.Linear(hid_dim, hid_dim), nn.Identity(), nn.ReLU(), nn.Identity()]
        layers += [nn.Linear(hid_dim, out_dim)]
        self.module = nn.Sequential(*layers)
        self._packed = None

    def __getstate__(self):   # (device copies of the weights are a cache)
        state = self.__dict__.copy()
        state["_packed"] = None
        return state

    def _pack(self, dev):
        """Padded device copies of the Linear layers, cached per device / parameter version: fp32 [Npad8][Kpad8]
        (fp16 [Npad128][Kpad64] for the opt-in fp16 path) weights + fp32 biases."""
        lin = [m for m in self.module if isinstance(m, nn.Linear)]
        key = (str(dev), self.precision) + tuple((m.weight._version, m.bias._version, m.weight.data_ptr()) for m in lin)
        if self._packed is None or self._packed[0] != key:
            half = self.precision == "fp16"
            gn, gk, wt = (128, 64, torch.float16) if half else (8, 8, torch.float32)
            packs = []
            for m in lin:
                n, k = m.weight.shape
                npad, kpad = -(-n // gn) * gn, -(-k // gk) * gk
                w = torch.zeros((npad, kpad), dtype=wt, device=dev)
                w[:n, :k] = m.weight.detach().to(dev, wt)
                b = torch.zeros(npad, dtype=torch.float32, device=dev)
                b[:n] = m.bias.detach().to(dev, torch.float32)
                packs.append((w, b, n, k, npad, kpad))
            self._packed = (key, packs)
        return self._packed[1]

    def forward(self, X):
        shape = X.shape
        X2 = X.reshape(-1, shape[-1])
        if not X2.is_cuda:
            return self.module(X2.float()).reshape(*shape[:-1], -1)
        L, dev = _lib.lib(), X2.device
        packs = self._pack(dev)
        rows = X2.shape[0]
        half = self.precision == "fp16"
        at = torch.float16 if half else torch.float32
        a = torch.zeros((rows, packs[0][5]), dtype=at, device=dev)
        a[:, :packs[0][3]] = X2.to(at)
        for i, (w, b, n, k, npad, kpad) in enumerate(packs):
            last = i == len(packs) - 1
            c = torch.empty((rows, npad), dtype=at, device=dev)
            if half:
                rc = L.lla_gemm_f16_ex(_lib.ptr(a), a.shape[1], _lib.ptr(w), _lib.ptr(b), _lib.ptr(c), npad, None, 0,
                                       rows, npad, kpad, _lib.LLA_EPI_F16 if last else _lib.LLA_EPI_RELU_F16,
                                       _lib.stream_ptr(dev))
                _lib.check(rc, "lla_gemm_f16_ex")
            else:
                # (a's width is the previous layer's Npad: a multiple of 8 >= this layer's K, padding columns 0)
                rc = L.lla_gemm_f32(_lib.ptr(a), a.shape[1], _lib.ptr(w), kpad, _lib.ptr(b), _lib.ptr(c), npad,
                                    rows, npad, kpad, 0 if last else 1, _lib.stream_ptr(dev))
                _lib.check(rc, "lla_gemm_f32")
            a = c          # (Npad is a valid Kpad of the next layer; padding columns are 0: zero weights, zero bias)
        out_dim = packs[-1][2]
        return a[:, :out_dim].float().reshape(*shape[:-1], out_dim)


class HRateHyperprior(HRateEstimator):
    """Scale-hyperprior coder twin (``lossyless/rates.py:572-756``), inference time.

    ``compress(z)`` -> ``[z_strings, side_z_strings]``: the side information ``side_encoder(z_in)``
    goes through the factorized ``EntropyBottleneck`` (HIP, one row per lane, table row = channel),
    its decoded value drives ``z_encoder`` -> (scales, means) -> ``build_indexes``, and ``z_in`` is
    coded by ``GaussianConditional`` with those per-element table rows (HIP,
    ``lla_rans_encode_indexed``).  ``get_indexes_means_hat`` reproduces the reference verbatim,
    including that it passes the *scales* as ``means`` (rates.py:694-696 overwrite ``means_hat``
    with ``atleast_ndim(scales_hat, 4)``), while ``forward_help`` -- like the reference's, :631-678 -- evaluates the
    likelihood with the predicted means.  The two MLPs run in fp32 (``mlp_precision="fp16"`` opts in to the fp16
    GEMMs: strings are then only decodable by a module with the same setting).
    """

    def __init__(self, z_dim, factor_dim=5, side_z_dim=None, is_pred_mean=True, mlp_precision="fp32", **kwargs):
        from .entropy import GaussianConditional
        super().__init__(z_dim, **kwargs)
        if side_z_dim is None:
            side_z_dim = max(10, z_dim // factor_dim)
        self.side_z_dim = side_z_dim
        self.is_pred_mean = is_pred_mean
        self.entropy_bottleneck = EntropyBottleneck(side_z_dim, **self.kwargs_ent_bottleneck)
        self.gaussian_conditional = GaussianConditional(None)
        self.mlp_precision = mlp_precision
        self.side_encoder, self.z_encoder = self.get_encoders()

    def get_encoders(self):
        """rates.py:617-629."""
        kwargs_mlp = dict(n_hid_layers=2, hid_dim=max(self.z_dim, 256), precision=self.mlp_precision)
        side_encoder = MLP(self.z_dim, self.side_z_dim, **kwargs_mlp)
        z_encoder = MLP(self.side_z_dim, self.z_dim * (2 if self.is_pred_mean else 1), **kwargs_mlp)
        return side_encoder, z_encoder

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        try:
            update_registered_buffers(self.gaussian_conditional, f"{prefix}gaussian_conditional",
                                      ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"],
                                      state_dict, policy="resize")
        except KeyError:
            pass
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward_help(self, z, _, __=None):
        """rates.py:631-678 (eval mode) -> ``(z_hat, -log q(z, s) per example, logs, other)`` with logs
        ``H_q_ZlS``, ``H_q_Z`` (= H_q_ZS, the reference's naming), ``H_q_S``, ``H_ZlX`` (+ the real rate's)."""
        nats_to_bits = 1.0 / math.log(BASE_LOG)

        def code_length(likelihood):           # -log q, summed over the latent dimension: [batch]
            return torch.log(likelihood).sum(-1).neg()

        z_in = self.process_z_in(z)
        # hyper-latent s through the factorized bottleneck, then the conditional model of z given s-hat
        s_hat, q_s = self.entropy_bottleneck(self.side_encoder(z_in))
        scales_hat, means_hat = self.chunk_params(self.z_encoder(s_hat))
        z_hat, q_z_given_s = self.gaussian_conditional(z_in, scales_hat, means=means_hat)
        len_s, len_z_given_s = code_length(q_s), code_length(q_z_given_s)
        len_joint = len_s + len_z_given_s
        # (the reference logs the joint length under the key H_q_Z, rates.py:660-661: kept, the evaluator reads it)
        logs = {"H_q_ZlS": len_z_given_s.mean() * nats_to_bits, "H_q_Z": len_joint.mean() * nats_to_bits,
                "H_q_S": len_s.mean() * nats_to_bits, "H_ZlX": 0}
        self._add_real_rate(z, logs)
        return self.process_z_out(z_hat), len_joint, logs, {}

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
