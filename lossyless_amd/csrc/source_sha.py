#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the kernel sources -- csrc/*.{hip,h,cpp,inc}, csrc/ablation/* + include/lossyless_amd.h -- that a
built liblossyless_amd*.so belongs to.  The Makefile compiles it into the library (lla_source_sha()), lossyless_amd/_lib.py
recomputes it from the tree and refuses a library built from other sources (the .so files are git-ignored and travel
prebuilt to the GPU box: a stale one would otherwise pass every test silently).  No imports beyond the stdlib."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
EXTS = ("hip", "h", "cpp", "inc")


def source_sha(csrc=HERE):
    h = hashlib.sha256()
    header = os.path.join(csrc, "..", "..", "include", "lossyless_amd.h")
    abl = os.path.join(csrc, "ablation")       # (the tools/ builds' launchers and switch readers: one sha for every build)
    files = [os.path.join(csrc, fn) for fn in os.listdir(csrc)]
    files += [os.path.join(abl, fn) for fn in os.listdir(abl)] if os.path.isdir(abl) else []
    for path in sorted(files, key=lambda q: os.path.relpath(q, csrc)) + [header]:
        if os.path.isfile(path) and path.rsplit(".", 1)[-1] in EXTS:
            with open(path, "rb") as f:
                h.update(os.path.relpath(path, csrc).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(source_sha())
