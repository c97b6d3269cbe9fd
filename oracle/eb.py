"""A11 / A15 / A4 / A5: EntropyBottleneck table derivation and the fp32 affine (test infrastructure).

Restates compressai==1.1.5 ``EntropyBottleneck.update()`` / ``_logits_cumulative`` /
``_pmf_to_cdf`` as reached from hub/compressor.py:63 (twin: lossyless/rates.py:286-305).
``compressai`` is not vendored in /root/reference; the recipe is SURVEY.md section 8(a)
row A11.  Two evaluations are offered:

* ``derive_tables(sd, "fp32")`` -- torch-CPU fp32, the arithmetic the reference itself
  runs (``update()`` executes before ``.to(device)``, hub/compressor.py:63-66);
* ``derive_tables(sd, "fp64")`` -- numpy float64, an independent evaluation used to
  bound how many 16-bit table entries sit on a rounding edge.

The integer step A12 goes through ``oracle.cbind.pmf_to_quantized_cdf``.
"""
import numpy as np
import torch

from . import cbind

FILTERS = (3, 3, 3, 3)
PREFIX = "entropy_bottleneck."


def _logits_fp32(sd, x):
    """x: torch fp32 [C,1,n] -> logits [C,1,n]; softplus(matrix) @ x + bias, tanh gate."""
    import torch.nn.functional as F
    for i in range(len(FILTERS) + 1):
        x = torch.matmul(F.softplus(sd[PREFIX + "_matrix%d" % i]), x)
        x = x + sd[PREFIX + "_bias%d" % i]
        if i < len(FILTERS):
            x = x + torch.tanh(sd[PREFIX + "_factor%d" % i]) * torch.tanh(x)
    return x


def _logits_fp64(sd, x):
    for i in range(len(FILTERS) + 1):
        m = sd[PREFIX + "_matrix%d" % i].double().numpy()
        m = np.logaddexp(0.0, m)
        x = np.matmul(m, x) + sd[PREFIX + "_bias%d" % i].double().numpy()
        if i < len(FILTERS):
            x = x + np.tanh(sd[PREFIX + "_factor%d" % i].double().numpy()) * np.tanh(x)
    return x


def derive_tables(sd, precision="fp32"):
    """state-dict -> dict(cdf int32 [C,W], cdf_len int32 [C], offset int32 [C],
    median fp32 [C], exp_scale fp32 [C], bias fp32 [C], pmf)."""
    q = sd[PREFIX + "quantiles"].float()
    med = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(med - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - med).int(), min=0)
    offset = -minima
    pmf_start = med - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]  # fp32 [C,1,L]

    if precision == "fp32":
        lower = _logits_fp32(sd, samples - 0.5)
        upper = _logits_fp32(sd, samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = (torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:]))
        pmf, tail = pmf.numpy(), tail.numpy()
    else:
        s64 = samples.double().numpy()
        lower = _logits_fp64(sd, s64 - 0.5)
        upper = _logits_fp64(sd, s64 + 0.5)
        sign = -np.sign(lower + upper)
        sig = lambda t: 1.0 / (1.0 + np.exp(-t))
        pmf = np.abs(sig(sign * upper) - sig(sign * lower))[:, 0, :]
        tail = sig(lower[:, 0, :1]) + sig(-upper[:, 0, -1:])

    C = q.shape[0]
    W = max_length + 2
    cdf = np.zeros((C, W), dtype=np.int32)
    for c in range(C):
        n = int(pmf_length[c])
        prob = np.concatenate([pmf[c, :n], tail[c]]).astype(np.float32)
        row = cbind.pmf_to_quantized_cdf(prob, 16)
        cdf[c, : n + 2] = row.astype(np.int32)
    return dict(
        cdf=cdf,
        cdf_len=(pmf_length + 2).numpy().astype(np.int32),
        offset=offset.numpy().astype(np.int32),
        median=med.numpy().astype(np.float32),
        # exp evaluated in float64 then rounded once: independent of libm / vector width
        exp_scale=torch.exp(sd["scaling"].double()).numpy().astype(np.float32),
        bias=sd["biasing"].float().numpy().astype(np.float32),
        pmf=np.asarray(pmf, dtype=np.float64),
        tail=np.asarray(tail, dtype=np.float64),
    )


def process_z_in(z, tab):
    """A4, hub/compressor.py:105-109: (z.float() + biasing) * exp(scaling), fp32 per op."""
    z = np.asarray(z, dtype=np.float32)
    return ((z + tab["bias"]).astype(np.float32) * tab["exp_scale"]).astype(np.float32)


def symbols_of(z, tab):
    """A13 front half: round_half_even(z_in - median).int()."""
    return cbind.quantise(np.asarray(z, dtype=np.float32), tab["bias"], tab["exp_scale"],
                          tab["median"])


def dequantise(symbols, tab):
    """A14 back half + A5 (hub/compressor.py:111-115): (float(sym) + median) / exp_scale - bias."""
    s = np.asarray(symbols).astype(np.float32)
    z_hat = (s + tab["median"]).astype(np.float32)
    return ((z_hat / tab["exp_scale"]).astype(np.float32) - tab["bias"]).astype(np.float32)


def represent(z, tab):
    """A15 + A5, hub/compressor.py:100-101: z -> z_hat without coding."""
    return dequantise(symbols_of(z, tab), tab)


def model_entropy_bits(tab):
    """sum_c H(quantised pmf_c) in bits -- SURVEY.md section 9.2 column."""
    total = 0.0
    for c in range(tab["cdf"].shape[0]):
        n = int(tab["cdf_len"][c])
        f = np.diff(tab["cdf"][c, :n].astype(np.float64)) / 65536.0
        f = f[f > 0]
        total += float(-(f * np.log2(f)).sum())
    return total
