"""CPU: pin the oracle (oracle/) -- hand-derived known answers, two independent
restatements agreeing, golden fixtures, algebraic properties.  No GPU, no product code."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import BETAS, GOLDEN, load_tables, sample_symbols
from oracle import cbind, container, eb, pyrans


# ---------------------------------------------------------------- A12 known answers
def test_pmf_to_cdf_exact_powers_of_two():
    # 0.5/0.25/0.25 at 16 bits: 32768/16384/16384, total already 65536
    assert cbind.pmf_to_quantized_cdf([0.5, 0.25, 0.25]).tolist() == [0, 32768, 49152, 65536]


def test_pmf_to_cdf_zero_bin_steals_from_smallest_donor():
    # p = [1-3e-6, 1e-6, 1e-6, 1e-6]: round(p*65536) = [65536,0,0,0] -> three empty bins.
    # Each takes one count from the only bin wider than 1 (bin 0), shifting the edges
    # between donor and taker down: hand result [0, 65533, 65534, 65535, 65536].
    got = cbind.pmf_to_quantized_cdf([1 - 3e-6, 1e-6, 1e-6, 1e-6]).tolist()
    assert got == [0, 65533, 65534, 65535, 65536]


def test_pmf_to_cdf_donor_to_the_right():
    # empty first bin, donor on its right: edges strictly between move UP by one
    got = cbind.pmf_to_quantized_cdf([1e-7, 0.5, 0.5]).tolist()
    assert got == [0, 1, 32768, 65536]


def test_pmf_to_cdf_renormalises_by_integer_total():
    # round(.3*65536)=19661 three times = 58983 total; each -> floor(65536*19661/58983)=21845
    got = cbind.pmf_to_quantized_cdf([0.3, 0.3, 0.3]).tolist()
    assert got == [0, 21845, 43690, 65536]


def test_pmf_to_cdf_picks_first_smallest_donor_on_ties():
    # bins: [A=2 counts, empty, B=2 counts, big]; smallest donors A and B tie -> first (A)
    p = [2 / 65536, 0.0, 2 / 65536, 1 - 4 / 65536]
    got = cbind.pmf_to_quantized_cdf(p).tolist()
    assert got == [0, 1, 2, 4, 65536]


# ---------------------------------------------------------------- A13 known answers
TOY = dict(cdf=np.array([[0, 32768, 65536]], np.int32), cdf_len=np.array([3], np.int32),
           offset=np.array([0], np.int32))


def _enc(sym, t=TOY):
    return cbind.rans_encode(np.asarray(sym, np.int32), t["cdf"], t["cdf_len"], t["offset"])


def test_rans_single_regular_symbol_by_hand():
    # x0 = 2^31, freq 2^15, start 0: limit 2^62 not reached;
    # x = ((2^31 / 2^15) << 16) + 0 + 0 = 2^32 -> flush words [lo=0, hi=1], little endian
    assert _enc([0]) == struct.pack("<II", 0, 1)


def test_rans_escape_with_zero_payload_by_hand():
    # sym 1 == escape index 1 -> raw = 2*(1-1) = 0 -> 0 digits: items [esc][digit 0].
    # reverse: digit 0: x = 2^31 << 4 = 2^35; esc: x = ((2^35/2^15)<<16) + 32768 = 2^36 + 2^15
    assert _enc([1]) == struct.pack("<II", 0x8000, 0x10)


def test_rans_negative_escape_by_hand():
    # sym -1: raw = -2*(-1)-1 = 1 -> 1 digit.  items: [esc][count=1][digit 1]
    # reverse: digit 1: x = 2^35 | 1; count 1: x = 2^39 | 0x11;
    # esc (freq 2^15, start 2^15): x = ((x >> 15) << 16) + (x & 0x7fff) + 0x8000
    x = ((1 << 35) | 1)
    x = (x << 4) | 1
    x = ((x >> 15) << 16) + (x & 0x7FFF) + 0x8000
    assert _enc([-1]) == struct.pack("<II", x & 0xFFFFFFFF, x >> 32)
    assert cbind.rans_decode(_enc([-1]), 1, TOY["cdf"], TOY["cdf_len"], TOY["offset"]).tolist() == [-1]


def test_rans_renormalisation_emits_word_by_hand():
    # 40 escapes-free symbols "0" with freq 2^15 add 1 bit each: x = 2^31 * 2^k until
    # x >= 2^62 triggers a 32-bit word.  After 31 symbols x = 2^62 -> 32nd put emits
    # word 0 and continues from 2^30 * 2 = 2^31... closed form below.
    t = dict(cdf=np.tile(np.array([[0, 32768, 65536]], np.int32), (40, 1)),
             cdf_len=np.full(40, 3, np.int32), offset=np.zeros(40, np.int32))
    x, words = 1 << 31, []
    for _ in range(40):
        if x >= (1 << 62):
            words.append(x & 0xFFFFFFFF)
            x >>= 32
        x = ((x >> 15) << 16) + (x & 0x7FFF)
    want = struct.pack("<%dI" % (2 + len(words)), x & 0xFFFFFFFF, x >> 32, *words[::-1])
    assert len(words) == 1
    assert _enc([0] * 40, t) == want


def test_stream_length_is_word_aligned_and_at_least_flush(tables_b005):
    sym = sample_symbols(tables_b005, 16, seed=3)
    for s in sym:
        b = cbind.rans_encode(s, tables_b005["cdf"], tables_b005["cdf_len"], tables_b005["offset"])
        assert len(b) % 4 == 0 and len(b) >= 8


# ---------------------------------------------------------------- two restatements agree
def test_c_and_python_restatements_agree(tables):
    sym = sample_symbols(tables, 6, seed=11, escape_boost=0.02)
    sym[0, :6] = [2 ** 20, -2 ** 20, 2 ** 29, -2 ** 29, 0, -1]
    for s in sym:
        a = cbind.rans_encode(s, tables["cdf"], tables["cdf_len"], tables["offset"])
        b = pyrans.encode(s, tables["cdf"], tables["cdf_len"], tables["offset"])
        assert a == b
        assert pyrans.decode(a, len(s), tables["cdf"], tables["cdf_len"], tables["offset"]) == s.tolist()
        assert np.array_equal(
            cbind.rans_decode(a, len(s), tables["cdf"], tables["cdf_len"], tables["offset"]), s)


def test_round_trip_with_heavy_escapes(tables):
    sym = sample_symbols(tables, 200, seed=5, escape_boost=0.3)
    pay, off = cbind.rans_encode_batch(sym, tables["cdf"], tables["cdf_len"], tables["offset"])
    back = cbind.rans_decode_batch(pay, off, sym.shape[1], tables["cdf"], tables["cdf_len"],
                                   tables["offset"])
    assert np.array_equal(back, sym)


# ---------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("tag", BETAS)
def test_golden_container(tag):
    tab = load_tables(tag)
    sym = np.load(os.path.join(GOLDEN, f"symbols_{tag}.npy"))
    strings = [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"]) for s in sym]
    with open(os.path.join(GOLDEN, f"golden_{tag}.bin"), "rb") as f:
        assert container.container_bytes(strings) == f.read()
    got = container.read_container(os.path.join(GOLDEN, f"golden_{tag}.bin"))
    assert got == strings


def test_container_layout_by_hand(tmp_path):
    p = tmp_path / "c.bin"
    container.write_container(p, [b"abcd", b"", b"xy"])
    assert p.read_bytes() == (b"\0\0\0\3" + b"\0\0\0\4abcd" + b"\0\0\0\0" + b"\0\0\0\2xy")


# ---------------------------------------------------------------- table facts
def test_tables_match_survey_facts():
    with open(os.path.join(GOLDEN, "tables_manifest.json")) as f:
        man = json.load(f)
    # W and the cdf_len / offset ranges per checkpoint
    assert man["1e-01"]["W"] == 32 and man["5e-02"]["W"] == 32 and man["1e-02"]["W"] == 33
    assert man["1e-01"]["cdf_len"] == [20, 32] and man["1e-01"]["offset"] == [-16, -8]
    assert man["5e-02"]["cdf_len"] == [23, 32] and man["5e-02"]["offset"] == [-18, -10]
    assert man["1e-02"]["cdf_len"] == [31, 33] and man["1e-02"]["offset"] == [-19, -11]


def test_tables_are_valid_cdfs(tables):
    for c in range(tables["cdf"].shape[0]):
        n = tables["cdf_len"][c]
        row = tables["cdf"][c, :n]
        assert row[0] == 0 and row[-1] == 65536 and (np.diff(row) > 0).all()
        assert (tables["cdf"][c, n:] == 0).all()


def test_model_entropy_in_reported_band(tables_b005):
    # reference reports 1506.6 bits/img on STL10 for b005 (notebooks/Hub.ipynb:253);
    # the model's own entropy must sit below it by the container/flush overhead
    h = eb.model_entropy_bits(tables_b005)
    assert 1300 < h < 1450


def test_quantise_is_round_half_even():
    tab = dict(bias=np.zeros(4, np.float32), exp_scale=np.ones(4, np.float32),
               median=np.zeros(4, np.float32))
    z = np.array([[0.5, 1.5, 2.5, -0.5]], np.float32)
    assert eb.symbols_of(z, tab).tolist() == [[0, 2, 2, 0]]
