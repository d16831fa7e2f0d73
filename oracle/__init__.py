"""TEST INFRASTRUCTURE ONLY — CPU oracle for the ScorePerformer hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker.  ``scoreperformer_amd`` never imports this package.

Parity status: PINNED.  ``oracle/ref_cpu.py`` (a functional fp32 PyTorch-CPU
restatement of the reference algorithm, citing reference file:line per function)
is checked in ``tests/test_oracle_golden.py`` against fixtures under
``tests/golden/`` that were produced by running the *reference itself*
(``/root/reference``, imported in the authoring container through
``oracle/refimport/make_golden.py``) on the same seeded inputs and weights.
"""
