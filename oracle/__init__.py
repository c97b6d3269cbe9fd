"""CPU oracle for the compress_dataset hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  ``lossyless_amd`` never does; the product
fails loudly when its HIP library is missing instead of routing through here.

What is restated and from where:

* ``rans_oracle.c``  -- A12 / A13 / A14 (pmf->cdf, rANS encode / decode), plain C.
* ``cbind.py``       -- ctypes view of that C library.
* ``pyrans.py``      -- the same A13 / A14 again as pure-Python big-int loops, used
  on small cases to cross-check the C transcription.
* ``eb.py``          -- A11 / A15 / A4 / A5: EntropyBottleneck table derivation and the
  fp32 affine (hub/compressor.py:56-63,95-115).
* ``vit.py``         -- A10: CLIP ViT-B/32 visual tower forward in fp32 torch-CPU ops.
* ``container.py``   -- A8: the ``.bin`` record format (hub/compressor.py:258-275).
* ``gc.py``          -- SURVEY.md 8(f) rank 4: GaussianConditional scale-table CDFs, ``build_indexes``
  and strings with a table row per symbol (lossyless/rates.py:567-729).

PARITY UNPINNED against live third-party code: ``compressai==1.1.5`` and
``clip==1.0`` (requirements/environment.yaml:98,105) hold the arithmetic and are
neither vendored in /root/reference nor importable here, and the reference has no
unit tests or golden vectors for this path.  What pins
the restatement instead: hand-derived known-answer streams
(tests/test_oracle.py), two independent restatements agreeing (C vs
pure-Python, fp32 vs fp64 table derivation), checkpoint-side invariants
(SURVEY.md 8c "evidence"), and the reference's recorded aggregate rate
(notebooks/Hub.ipynb:253) as a plausibility band.
"""
