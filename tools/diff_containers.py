#!/usr/bin/env python3
"""Diff two `.bin` containers (hub/compressor.py:192-196 layout) record by record and classify every difference.

SURVEY.md 8(e): the file of an N-rank run must equal the 1-rank file.  When two files differ this says WHERE:
which records (index, the shard of an n-rank run they fall in, distance to the nearest shard boundary), whether
the record lengths differ, and -- after decoding both versions of every differing record with the library's HOST
coder (needs no GPU) -- how far apart the two embeddings are in quantisation steps (symbols) per dimension.

  python tools/diff_containers.py one.bin two.bin [--ranks 2] [--rate b005] [--max 64]

Prints one JSON object; exit status 0 when the files are equal, 1 when they differ.  Importable:
`diff_containers(a, b, ranks=..)` returns the same dict (tests/test_gpu_configs.py calls it on a mismatch).
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _index(blob):
    from lossyless_amd import _lib
    L = _lib.lib()
    n = ctypes.c_uint32(0)
    _lib.check(L.lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size, None, 0, ctypes.byref(n)),
               "lla_container_index")
    off = np.zeros(int(n.value) + 1, dtype=np.uint64)
    _lib.check(L.lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                     off.ctypes.data_as(ctypes.c_void_p), off.size, ctypes.byref(n)),
               "lla_container_index")
    return int(n.value), off.astype(np.int64)


def _decode(comp, body, off, idx):
    """fp32 [len(idx), 512] embeddings of records `idx` (host coder)."""
    out = np.empty((len(idx), comp.z_dim), dtype=np.float32)
    for k, i in enumerate(idx):
        rec = body[off[i]:off[i + 1]]
        out[k] = comp._decode_records_host(rec, np.array([0, rec.size], dtype=np.uint64), 1)[0]
    return out


def diff_containers(path_a, path_b, ranks=1, rate="b005", max_records=64):
    a = np.fromfile(path_a, dtype=np.uint8)
    b = np.fromfile(path_b, dtype=np.uint8)
    na, offa = _index(a)
    nb, offb = _index(b)
    res = dict(a=str(path_a), b=str(path_b), records_a=na, records_b=nb, bytes_a=int(a.size), bytes_b=int(b.size))
    if na != nb:
        res.update(equal=False, verdict="record counts differ")
        return res
    body_a, body_b = a[4:], b[4:]
    same_layout = bool(np.array_equal(offa, offb))
    if same_layout:
        pos = np.nonzero(body_a != body_b)[0]
        bad = np.unique(np.searchsorted(offa, pos, side="right") - 1)
    else:
        lens_differ = np.nonzero(np.diff(offa) != np.diff(offb))[0]
        bad = set(int(i) for i in lens_differ)
        # records of equal length but (possibly) shifted position: compare content
        eq_len = np.nonzero(np.diff(offa) == np.diff(offb))[0]
        for i in eq_len:
            if not np.array_equal(body_a[offa[i]:offa[i + 1]], body_b[offb[i]:offb[i + 1]]):
                bad.add(int(i))
        bad = np.array(sorted(bad), dtype=np.int64)
    res.update(equal=bad.size == 0, same_record_lengths=same_layout, differing_records=int(bad.size))
    if bad.size == 0:
        res["verdict"] = "equal"
        return res
    # shards of an n-rank run (distributed.shard_bounds: contiguous, the first N % W ranks one image longer)
    from lossyless_amd.distributed import shard_bounds
    bounds = [shard_bounds(na, r, ranks) for r in range(ranks)]
    starts = np.array([lo for lo, _ in bounds] + [na], dtype=np.int64)
    import hubconf
    comp, _ = getattr(hubconf, f"clip_compressor_{rate}")(device="cpu", clip_weights="synthetic")
    step = np.exp(-comp.scaling.detach().cpu().numpy().astype(np.float64))   # one symbol in embedding units, per dimension
    rows = []
    shown = bad[:max_records]
    za, zb = _decode(comp, body_a, offa, shown), _decode(comp, body_b, offb, shown)
    for k, i in enumerate(shown):
        d = (za[k].astype(np.float64) - zb[k].astype(np.float64)) / step
        shard = int(np.searchsorted(starts, i, side="right") - 1)
        rows.append(dict(record=int(i), shard=shard, offset_in_shard=int(i - starts[shard]),
                         to_shard_end=int(starts[shard + 1] - 1 - i),
                         len_a=int(offa[i + 1] - offa[i]) - 4, len_b=int(offb[i + 1] - offb[i]) - 4,
                         dims_differing=int(np.count_nonzero(d)), max_symbol_delta=float(np.abs(d).max()),
                         max_abs_embedding_delta=float(np.abs(za[k] - zb[k]).max())))
    res["records"] = rows
    worst = max(r["max_symbol_delta"] for r in rows)
    res["max_symbol_delta"] = worst
    res["verdict"] = (f"{bad.size} of {na} records differ; every decoded difference is <= {worst:g} quantisation step(s): "
                      "the embeddings differ in the last bits (tower), the coder and the gather are intact"
                      if worst <= 2 else f"{bad.size} of {na} records differ by up to {worst:g} steps: NOT a rounding-level difference")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--rate", default="b005", choices=["b005", "b001", "b01"])
    ap.add_argument("--max", type=int, default=64)
    args = ap.parse_args()
    res = diff_containers(args.a, args.b, args.ranks, args.rate, args.max)
    print(json.dumps(res))
    return 0 if res["equal"] else 1


if __name__ == "__main__":
    sys.exit(main())
