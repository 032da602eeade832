"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the MuRaL hot path (SURVEY.md
section 8a).  Nothing in ``mural_amd/`` (the product) may import, call, link or
execute anything in this package: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` use it, and there only as the
*checker*, never as the thing being measured or shipped.

Pinning status: the reference (CaiLiLab/MuRaL) ships no tests or golden vectors
for this path, so the oracle is pinned against outputs of
the reference itself, imported in the build container by
``oracle/make_golden.py`` (stub recipe in ``oracle/ref_import.py``); the
resulting vectors are committed under ``tests/golden/`` and
``tests/test_oracle_golden.py`` re-checks the oracle against them on every run.
"""
