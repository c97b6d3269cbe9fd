"""The training-side twin (lossyless_amd/rates.py) driven the way the reference's evaluator drives its rate
estimator -- lossyless/learnable_compressors.py:84 (make_pickable_ after construction), :123-177 (forward:
``rate_estimator.compress(z, self)`` or ``rate_estimator(z, p_Zlx, self)``), :339-341 (on_test_epoch_start ->
prepare_compressor_), :436 (make_pickable_ again) -- on the CPU, where the twin codes with the library's host
coder.  The stand-in module below is written for this test; it is not the reference's file."""
import math
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import ROOT, load_tables
from lossyless_amd.rates import HRateFactorizedPrior, HRateHyperprior
from oracle import cbind, eb


class _Delta:
    """Deterministic p(Z|x) as the evaluator hands it over (lossyless/distributions.py:139-152)."""

    def __init__(self, loc):
        self.loc = loc

    def rsample(self):
        return self.loc

    def detach(self, is_grad_flow=False):
        return _Delta(self.loc.detach())


class _StandInCompressor(torch.nn.Module):
    """What LearnableCompressor does with its rate estimator, and nothing else."""

    def __init__(self, rate_estimator):
        super().__init__()
        self.current_epoch = 0
        self.rate_estimator = rate_estimator
        self.rate_estimator.make_pickable_()            # learnable_compressors.py:84

    def forward(self, z, is_compress=False):            # :155-165 (the encoder is the identity here)
        p_Zlx = _Delta(z)
        z = p_Zlx.rsample()
        if is_compress:
            return self.rate_estimator.compress(z, self)
        z_hat, rates, r_logs, r_other = self.rate_estimator(z, p_Zlx, self)
        return z_hat, rates, r_logs, r_other

    def on_test_epoch_start(self):                      # :339-341
        self.rate_estimator.prepare_compressor_()

    def set_featurizer(self):                           # :427-436
        self.eval()
        self.rate_estimator.make_pickable_()


def _likelihood_fp64(eb_module, v):
    """Independent evaluation of the factorized density (Balle et al. 2018, appendix 6.1) in float64 numpy:
    p(v) = |sigmoid(s * c(v + .5)) - sigmoid(s * c(v - .5))| with c the cumulative logits, channel by channel."""
    P = {n: p.detach().double().numpy() for n, p in eb_module.named_parameters()}
    C = v.shape[1]
    out = np.empty_like(v, dtype=np.float64)

    def logits(x, c):
        h = x[None, :]                                   # [1, B]
        for i in range(5):
            h = np.log1p(np.exp(P[f"_matrix{i}"][c])) @ h + P[f"_bias{i}"][c]
            if i < 4:
                h = h + np.tanh(P[f"_factor{i}"][c]) * np.tanh(h)
        return h[0]

    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    for c in range(C):
        lo, up = logits(v[:, c] - 0.5, c), logits(v[:, c] + 0.5, c)
        s = -np.sign(lo + up)
        out[:, c] = np.abs(sig(s * up) - sig(s * lo))
    return out


def _factorized(tag="5e-02"):
    sd = torch.load(os.path.join(ROOT, "lossyless_amd", "assets", f"beta{tag}_factorized_rate.pt"),
                    map_location="cpu", weights_only=True)
    m = HRateFactorizedPrior(512, kwargs_ent_bottleneck=dict(init_scale=10, filters=[3, 3, 3, 3]))
    m.load_state_dict(sd)
    return m


def _model_like_z(m, B, seed):
    """Representations whose z_in sits a few bins around the medians (what the rate model was trained on)."""
    g = torch.Generator().manual_seed(seed)
    med = m.entropy_bottleneck._medians()
    z_in = med[None, :] + 1.5 * torch.randn(B, 512, generator=g)
    return (z_in / m.scaling.detach().exp() - m.biasing.detach()).float()


@torch.no_grad()   # (the evaluator's test loop runs without autograd)
def test_evaluator_protocol_on_the_factorized_twin():
    m = _factorized()
    lc = _StandInCompressor(m).eval()
    assert m.is_can_compress and m.is_coder_updated and not m.is_coder_present
    assert not m.is_compute_real_rate
    pickle.loads(pickle.dumps(lc))                       # what make_pickable_ is for (DDP spawns)
    z = _model_like_z(m, 64, 0)

    # coder detached: the forward still evaluates, without the real rate; compress refuses
    z_hat, rates, logs, other = lc(z)
    assert set(logs) == {"H_q_Z", "H_ZlX"} and other == {}
    assert rates.shape == (64,) and z_hat.shape == (64, 512)
    with pytest.raises(RuntimeError):
        lc(z, is_compress=True)

    lc.on_test_epoch_start()
    assert m.is_coder_present and m.is_compute_real_rate
    z_hat2, rates2, logs2, _ = lc(z)
    assert torch.equal(z_hat, z_hat2) and torch.equal(rates, rates2)
    assert {"H_q_Z", "H_ZlX", "n_bits", "compress_time", "receiver_time"} <= set(logs2)
    assert logs2["compress_time"] > 0 and logs2["receiver_time"] > 0

    # H_q_Z against an independent float64 evaluation of the density, and against the coder's real rate
    tab = load_tables("5e-02")
    sym = eb.symbols_of(z.numpy(), tab)
    v = sym.astype(np.float64) + tab["median"][None, :].astype(np.float64)
    lik = np.maximum(_likelihood_fp64(m.entropy_bottleneck, v), 1e-9)
    want_bits = float((-np.log2(lik)).sum(1).mean())
    assert abs(float(logs2["H_q_Z"]) - want_bits) <= 1e-5 * want_bits
    assert abs(float(rates2.mean()) / math.log(2) - float(logs2["H_q_Z"])) < 1e-3
    assert want_bits < logs2["n_bits"] < 1.05 * want_bits + 64     # rANS: entropy + flush + table quantisation

    # z_hat = dequantised symbols (bit-exact integer stage), strings = the oracle coder's on the same symbols
    assert np.array_equal(z_hat.numpy(), eb.dequantise(sym, tab))
    all_strings = lc(z, is_compress=True)
    assert isinstance(all_strings, list) and len(all_strings) == 1
    assert all_strings[0] == [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"]) for s in sym]
    assert torch.equal(m.decompress(all_strings), z_hat)
    assert logs2["n_bits"] == 8 * sum(map(len, all_strings[0])) / 64

    lc.set_featurizer()
    assert not m.is_coder_present
    pickle.loads(pickle.dumps(lc))


def test_warmup_and_disjoint_branches_recompute_the_rate_on_detached_inputs():
    m = _factorized()
    m.warmup_k_epoch = 3
    lc = _StandInCompressor(m).eval()
    z = _model_like_z(m, 8, 1).requires_grad_(True)
    _, rates, _, _ = lc(z)                               # epoch 0 < warm-up: rate of the detached representation
    assert rates.shape == (8,)
    lc.current_epoch = 5
    m.is_endToEnd = False
    _, rates_b, _, _ = lc(z)
    assert torch.equal(rates, rates_b)


@torch.no_grad()
def test_hyperprior_twin_forward_help_on_cpu():
    torch.manual_seed(0)
    m = HRateHyperprior(512).eval()
    assert type(m).__mro__[1].__name__ == "HRateEstimator"          # the reference's hierarchy (rates.py:572)
    lc = _StandInCompressor(m)
    z = torch.randn(16, 512, generator=torch.Generator().manual_seed(2))
    z_hat, rates, logs, other = lc(z)
    assert set(logs) == {"H_q_ZlS", "H_q_Z", "H_q_S", "H_ZlX"}     # tables not built, coder detached: no real rate
    assert rates.shape == (16,) and z_hat.shape == (16, 512) and bool(torch.isfinite(rates).all())
    assert abs(float(logs["H_q_Z"]) - float(logs["H_q_ZlS"]) - float(logs["H_q_S"])) < 1e-2
    assert abs(float(rates.mean()) / math.log(2) - float(logs["H_q_Z"])) < 1e-2
    pickle.loads(pickle.dumps(lc))
