"""Freeze the reference's three hub checkpoints into lossyless_amd/assets/ (run in the
build container, where /root/reference exists; the GPU box only ever sees the outputs).

For each hub/beta*/factorized_rate.pt (data written by the reference's utils/save_hub.py:46-50;
SURVEY.md F4) this loads the tensors and derives the integer tables with the ORACLE
(oracle/eb.py::derive_tables, fp32 torch-CPU + oracle C pmf_to_quantized_cdf: the arithmetic the
reference performs at load, hub/compressor.py:63).  Those tables are what is written to
tests/golden/tables_*.npz (+ SHA-256) -- the fixture comes from the checker, never from the
product -- and the script then REQUIRES the product's EntropyBottleneck.update() to reproduce
them exactly before it freezes them into the shipped state-dict
(`_quantized_cdf/_offset/_cdf_length` populated).  A state-dict that already carries tables
makes update() a no-op -- the reference's own mechanism -- so the integer tables stop depending
on the host's libm.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lossyless_amd.entropy import EntropyBottleneck, update_registered_buffers  # noqa: E402
from oracle import eb as oracle_eb  # noqa: E402

REF = "/root/reference/hub"
BETAS = {"1e-01": 0.1, "5e-02": 0.05, "1e-02": 0.01}


def main():
    os.makedirs(os.path.join(ROOT, "lossyless_amd", "assets"), exist_ok=True)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    manifest = {}
    for tag in BETAS:
        sd = torch.load(f"{REF}/beta{tag}/factorized_rate.pt", map_location="cpu", weights_only=True)
        eb = EntropyBottleneck(512, init_scale=10, filters=[3, 3, 3, 3])
        update_registered_buffers(eb, "entropy_bottleneck",
                                  ["_quantized_cdf", "_offset", "_cdf_length"], sd)
        eb.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items()
                            if k.startswith("entropy_bottleneck.")})
        ot = oracle_eb.derive_tables(sd, "fp32")
        tab = {k: np.ascontiguousarray(ot[k]) for k in ("cdf", "cdf_len", "offset", "median",
                                                         "exp_scale", "bias")}
        assert eb.update() is True
        assert np.array_equal(eb._quantized_cdf.numpy(), tab["cdf"]), "product update() != oracle"
        assert np.array_equal(eb._cdf_length.numpy(), tab["cdf_len"])
        assert np.array_equal(eb._offset.numpy(), tab["offset"])
        out = dict(sd)
        out["entropy_bottleneck._quantized_cdf"] = torch.from_numpy(tab["cdf"].astype(np.int32))
        out["entropy_bottleneck._offset"] = torch.from_numpy(tab["offset"].astype(np.int32))
        out["entropy_bottleneck._cdf_length"] = torch.from_numpy(tab["cdf_len"].astype(np.int32))
        path = os.path.join(ROOT, "lossyless_amd", "assets", f"beta{tag}_factorized_rate.pt")
        torch.save(out, path)
        npz = os.path.join(ROOT, "tests", "golden", f"tables_{tag}.npz")
        np.savez_compressed(npz, **tab)
        h = hashlib.sha256()
        for k in ("cdf", "cdf_len", "offset", "median", "exp_scale", "bias"):
            h.update(np.ascontiguousarray(tab[k]).tobytes())
        manifest[tag] = dict(sha256=h.hexdigest(), W=int(tab["cdf"].shape[1]),
                             cdf_len=[int(tab["cdf_len"].min()), int(tab["cdf_len"].max())],
                             offset=[int(tab["offset"].min()), int(tab["offset"].max())])
        print(tag, manifest[tag])
    with open(os.path.join(ROOT, "tests", "golden", "tables_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
