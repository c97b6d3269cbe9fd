"""A8: the `.bin` record format of hub/compressor.py:192-196,233-237,258-275 (test infrastructure).

big-endian u32 N, then N x { big-endian u32 len_i, len_i raw bytes }.
"""
import struct


def write_container(path, strings):
    with open(path, "wb") as f:
        f.write(struct.pack(">I", len(strings)))
        for s in strings:
            f.write(struct.pack(">I", len(s)))
            if len(s):
                f.write(s)


def container_bytes(strings):
    parts = [struct.pack(">I", len(strings))]
    for s in strings:
        parts.append(struct.pack(">I", len(s)))
        parts.append(bytes(s))
    return b"".join(parts)


def read_container(path):
    with open(path, "rb") as f:
        blob = f.read()
    (n,) = struct.unpack_from(">I", blob, 0)
    pos, out = 4, []
    for _ in range(n):
        (k,) = struct.unpack_from(">I", blob, pos)
        pos += 4
        out.append(blob[pos:pos + k])
        pos += k
    assert pos == len(blob)
    return out
