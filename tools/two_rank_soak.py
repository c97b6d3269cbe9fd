#!/usr/bin/env python3
"""Repeat the 1-rank-vs-2-rank file identity check of tests/test_gpu_configs.py K times and log
every run: one 1-rank file, then K two-rank (gloo, both ranks on the one GPU) runs compared with it; every mismatch
is classified by tools/diff_containers.py (which records, which shard, how many quantisation steps).

  python tools/two_rank_soak.py --runs 20 --images 1000000 --out gpurun_out/two_rank_soak.jsonl

The environment is passed through (LLA_LIB=<variant .so> selects a library build).  Exit status = number of
mismatching runs (0 = clean).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def bench(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LOSSYLESS_CLIP_WEIGHTS="synthetic")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", *args],
                       env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-1500:] + r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=20)
    ap.add_argument("--images", type=int, default=1_000_000)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--out", default="gpurun_out/two_rank_soak.jsonl")
    ap.add_argument("--tmp", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--cu-split", choices=["cu", "xcd", "none"], default=None,
                    help="bench.py --cu-split for the two-rank runs (default: bench.py gives ranks that share a GPU disjoint XCDs; "
                         "none = let them share every XCD)")
    args = ap.parse_args()
    from tools.diff_containers import diff_containers
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    one, two = os.path.join(args.tmp, "soak_one.bin"), os.path.join(args.tmp, "soak_two.bin")
    bad = 0
    with open(args.out, "a") as log:
        def emit(**kw):
            log.write(json.dumps(kw) + "\n")
            log.flush()
            print(json.dumps(kw), flush=True)
        t0 = time.time()
        r1 = bench("--gpus", "1", "--dataset-images", str(args.images), "--keep-file", one)
        emit(run="one_rank", sha=r1["file_sha256"], img_per_sec=r1["value"], lib=os.environ.get("LLA_LIB", "default"))
        for k in range(args.runs):
            r2 = bench("--gpus", str(args.ranks), "--backend", "gloo", "--dataset-images", str(args.images),
                       "--keep-file", two, *(["--cu-split", args.cu_split] if args.cu_split else []))
            same = r2["file_sha256"] == r1["file_sha256"]
            rec = dict(run=k, equal=same, img_per_sec=r2["value"], elapsed=round(time.time() - t0, 1))
            if not same:
                bad += 1
                rec["diff"] = diff_containers(one, two, ranks=args.ranks)
            emit(**rec)
        emit(run="summary", runs=args.runs, mismatching=bad, images=args.images, ranks=args.ranks)
    return bad


if __name__ == "__main__":
    sys.exit(main())
