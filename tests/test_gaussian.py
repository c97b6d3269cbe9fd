"""CPU: the GaussianConditional mirror's host logic (table derivation through the host C-ABI,
index building, argument checks) against the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import gc


@pytest.fixture(scope="module")
def tables():
    return gc.derive_tables(gc.get_scale_table())


@pytest.fixture(scope="module")
def cond():
    from lossyless_amd.entropy import GaussianConditional
    g = GaussianConditional(None)
    assert g.update_scale_table(gc.get_scale_table()) is True
    assert g.update_scale_table(gc.get_scale_table()) is False      # kept unless forced
    return g


def test_scale_table_matches_reference_recipe():
    from lossyless_amd.rates import get_scale_table
    st = get_scale_table()
    assert st.shape == (64,) and abs(float(st[0]) - 0.11) < 1e-6 and abs(float(st[-1]) - 256) < 1e-3
    assert torch.equal(st, gc.get_scale_table())


def test_tables_equal_oracle_and_are_valid_cdfs(cond, tables):
    assert np.array_equal(cond._quantized_cdf.numpy(), tables["cdf"])
    assert np.array_equal(cond._cdf_length.numpy(), tables["cdf_len"])
    assert np.array_equal(cond._offset.numpy(), tables["offset"])
    cdf, ln, off = tables["cdf"], tables["cdf_len"], tables["offset"]
    # smallest scale: 3 bins + tail; largest: 2*ceil(256 * 6.1) + 1 bins
    assert ln[0] == 5 and ln[-1] == cdf.shape[1] and (np.diff(ln) >= 0).all()
    assert np.array_equal(off, -(ln - 3) // 2)
    for i in range(cdf.shape[0]):
        row = cdf[i, :ln[i]]
        assert row[0] == 0 and row[-1] == 65536 and (np.diff(row) > 0).all()
        assert (cdf[i, ln[i]:] == 0).all()
        freq = np.diff(row)[:-1]
        assert freq[len(freq) // 2] >= freq.max() - 16            # mode at the centre (up to steals)


def test_float64_derivation_bounds_the_rounding_edge_cases(tables):
    """Same caveat as SURVEY.md F6 for the factorized tables: a float64 erfc moves a small share
    of the 16-bit edges by one or two counts -- tables must be shipped, not re-derived."""
    t64 = gc.derive_tables(gc.get_scale_table(), precision="fp64")
    assert np.array_equal(t64["cdf_len"], tables["cdf_len"])
    diff = np.abs(t64["cdf"].astype(np.int64) - tables["cdf"].astype(np.int64))
    assert diff.max() <= 4 and (diff > 0).mean() < 0.05


def test_build_indexes(cond, tables):
    st = tables["scale_table"]
    s = torch.tensor([0.0, 0.05, 0.11, 0.1100001, 0.5, 7.3, 255.9, 256.0, 1e4], dtype=torch.float32)
    got = cond.build_indexes(s).numpy()
    assert np.array_equal(got, gc.build_indexes(s.numpy(), st))
    assert got[0] == 0 and got[-1] == 63 and (np.diff(got) >= 0).all()
    # by definition: the first level >= max(scale, bound), capped at the last
    want = [min(int(np.searchsorted(st, max(np.float32(v), np.float32(0.11)), side="left")), 63)
            for v in s.numpy()]                                   # all in fp32, like the buffers
    assert got.tolist() == want
    g = torch.Generator().manual_seed(0)
    r = torch.exp(torch.randn(4, 7, 1, 1, generator=g) * 3)
    assert np.array_equal(cond.build_indexes(r).numpy(), gc.build_indexes(r.numpy(), st))


def test_constructor_checks():
    from lossyless_amd.entropy import GaussianConditional
    with pytest.raises(ValueError):
        GaussianConditional("x")
    with pytest.raises(ValueError):
        GaussianConditional([])
    with pytest.raises(ValueError):
        GaussianConditional([1.0, 0.5])
    with pytest.raises(ValueError):
        GaussianConditional(None, scale_bound=-1.0)
    g = GaussianConditional([0.5, 1.0, 2.0], scale_bound=None)
    assert float(g.scale_bound) == 0.5
    with pytest.raises(RuntimeError):
        g.device_tables()                      # update_scale_table() not called yet
    sd = g.state_dict()
    assert {"scale_table", "scale_bound", "_offset", "_quantized_cdf", "_cdf_length"} <= set(sd)


def test_oracle_roundtrip_with_escapes(tables):
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 64, size=(5, 41)).astype(np.int32)
    sym = np.rint(rng.normal(size=idx.shape) * tables["scale_table"][idx] * 1.5).astype(np.int32)
    sym[0, :6] = [10 ** 6, -10 ** 6, 2 ** 30, -(2 ** 30), 0, 1]     # far outside every table
    strings = gc.compress(sym, idx, tables)
    assert all(len(s) % 4 == 0 and len(s) >= 8 for s in strings)
    assert np.array_equal(gc.decompress(strings, idx, tables), sym)


def test_hyperprior_twin_state_dict_layout():
    from lossyless_amd.rates import HRateHyperprior
    m = HRateHyperprior(512)
    keys = set(m.state_dict())
    assert m.side_z_dim == 102
    for k in ("scaling", "biasing", "side_encoder.module.0.weight", "side_encoder.module.8.bias",
              "z_encoder.module.4.weight", "gaussian_conditional.scale_table",
              "gaussian_conditional._quantized_cdf", "entropy_bottleneck.quantiles"):
        assert k in keys, k
    assert m.side_encoder.module[0].weight.shape == (512, 512)
    assert m.side_encoder.module[8].weight.shape == (102, 512)
    assert m.z_encoder.module[8].weight.shape == (1024, 512)
    assert not m.is_coder_updated
    m.update(force=True)
    assert m.is_coder_updated
    # the dynamically sized buffers survive a state-dict round trip (rates.py:726-756 hook)
    m2 = HRateHyperprior(512)
    m2.load_state_dict(m.state_dict())
    assert m2.is_coder_updated
    assert torch.equal(m2.gaussian_conditional._quantized_cdf, m.gaussian_conditional._quantized_cdf)
    # ... keeping their dtypes: the scale table is float (compressai resizes the registered buffer in place; re-registering
    # it as int truncated the levels and a LOADED module built other indexes than the one that wrote the strings)
    assert m2.gaussian_conditional.scale_table.dtype == torch.float32
    assert torch.equal(m2.gaussian_conditional.scale_table, m.gaussian_conditional.scale_table)
    s = torch.rand(7, 512) * 3
    assert torch.equal(m2.gaussian_conditional.build_indexes(s), m.gaussian_conditional.build_indexes(s))


def test_committed_gaussian_fixture(tables):
    """tests/golden/gaussian_golden.npz (tools/make_golden.py): table digest and strings reproduce."""
    import hashlib
    import os
    from conftest import GOLDEN
    from oracle import container
    g = np.load(os.path.join(GOLDEN, "gaussian_golden.npz"))
    digest = hashlib.sha256(tables["cdf"].tobytes() + tables["cdf_len"].tobytes()
                            + tables["offset"].tobytes()).hexdigest()
    assert digest == str(g["table_sha256"])
    strings = gc.compress(g["symbols"], g["indexes"], tables)
    assert container.container_bytes(strings) == g["container"].tobytes()
    assert np.array_equal(gc.decompress(strings, g["indexes"], tables), g["symbols"])
