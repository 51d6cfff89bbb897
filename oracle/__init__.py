"""CPU oracle for the STEm-Seg embed+cluster hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``stem-seg_amd/``) may import, call or
link anything in this directory.  The only legitimate users are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the reported CPU
baseline, never as the thing measured or shipped.

The oracle is a restatement (plain PyTorch *CPU* fp32 ops for the floating-point stages, numpy for the
integer bookkeeping) of the reference algorithm; every function cites the reference file:line it
follows.  Parity status: PINNED -- each function is checked in ``tests/test_oracle_vs_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the reference itself
(``tools/make_goldens.py``; the reference ships no tests or golden vectors of its own, SURVEY.md
section 4 and 8(c)).
"""
