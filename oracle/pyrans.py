"""Second, independent restatement of A13 / A14 with Python big ints (test infrastructure).

Written from SURVEY.md section 8(a) rows A13/A14 without looking at rans_oracle.c's
control flow: it keeps an explicit forward list of "operations" and then runs the
coder over it backwards, so a transcription slip in either file shows up as a
disagreement in tests/test_oracle.py.  Small inputs only.
"""
import struct

LOW = 1 << 31
PROB = 16
NIB = 4


def _ops_for_symbol(sym, row, length, off):
    esc = length - 2
    v = sym - off
    raw = None
    if v < 0:
        raw, v = -2 * v - 1, esc
    elif v >= esc:
        raw, v = 2 * (v - esc), esc
    ops = [("sym", row[v], row[v + 1] - row[v])]
    if raw is not None:
        nd = 0
        while (raw >> (NIB * nd)) != 0:
            nd += 1
        left = nd
        while left >= 15:
            ops.append(("nib", 15))
            left -= 15
        ops.append(("nib", left))
        for d in range(nd):
            ops.append(("nib", (raw >> (NIB * d)) & 15))
    return ops


def encode(symbols, cdf, cdf_len, offset):
    ops = []
    for c, s in enumerate(symbols):
        ops += _ops_for_symbol(int(s), [int(t) for t in cdf[c]], int(cdf_len[c]), int(offset[c]))
    x = LOW
    words = []  # emitted in time order; the string is their reverse, after the 2 flush words
    for op in reversed(ops):
        if op[0] == "sym":
            _, start, freq = op
            if x >= ((LOW >> PROB) << 32) * freq:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // freq) << PROB) + (x % freq) + start
        else:
            if x >= ((LOW >> PROB) << 32) * (1 << (PROB - NIB)):
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = (x << NIB) | op[1]
    stream = [x & 0xFFFFFFFF, x >> 32] + words[::-1]
    return struct.pack("<%dI" % len(stream), *stream)


def decode(data, n, cdf, cdf_len, offset):
    words = list(struct.unpack("<%dI" % (len(data) // 4), data)) + [0, 0]
    x = words[0] | (words[1] << 32)
    pos = 2

    def nib():
        nonlocal x, pos
        d = x & 15
        x >>= NIB
        if x < LOW:
            x = (x << 32) | words[pos]
            pos += 1
        return d

    out = []
    for c in range(n):
        row = [int(t) for t in cdf[c]][: int(cdf_len[c])]
        esc = int(cdf_len[c]) - 2
        cf = x & 0xFFFF
        s = next(k for k, t in enumerate(row) if t > cf) - 1
        x = (row[s + 1] - row[s]) * (x >> PROB) + cf - row[s]
        if x < LOW:
            x = (x << 32) | words[pos]
            pos += 1
        v = s
        if s == esc:
            d = nib()
            nd = d
            while d == 15:
                d = nib()
                nd += d
            raw = 0
            for j in range(nd):
                raw |= nib() << (NIB * j)
            v = -(raw >> 1) - 1 if raw & 1 else (raw >> 1) + esc
        out.append(v + int(offset[c]))
    return out
