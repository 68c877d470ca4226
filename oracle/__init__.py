"""CPU oracle for the lceT5 training hot path and the paced negative selection.

TEST INFRASTRUCTURE ONLY.  Nothing under this package is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product package ``pacednegatives_amd`` never imports it and fails loudly
when its HIP library is missing.

Each function restates, in plain torch fp32 / numpy, one piece of the reference
(citations are ``file:line`` under the reference checkout) or of the third-party
T5 arithmetic the reference calls (Hugging Face ``modeling_t5.py``, cited as
``HF:line`` for transformers 5.15.0).

Pinning: the restatement is checked against fixtures under ``tests/golden/``
that were produced in the build container by ``tools/make_golden.py``, which
imports the reference's own ``pacednegatives`` modules and Hugging Face's T5
and records their outputs on seeded inputs.  The reference ships no tests or
golden vectors of its own.  Two pieces have no executable reference in the
container and are therefore "parity unpinned": ``transformers.AdamW`` (removed
upstream; restated from its published update rule) and Lightning's DDP
gradient averaging.
"""
