"""CPU oracle for the MARL hot path (TEST INFRASTRUCTURE - NOT PRODUCT CODE).

This package is a from-the-formulas CPU restatement (torch-CPU fp32 / numpy) of the
reference's hot path: RNNQNet agent step/unroll, VDN/QMIX/QPLEX/QTRAN mixers, the
QLearner / QTRANLearner train step (loss, clip-norm, RMSprop/Adam) and the rollout.
Every function cites the reference file:line it follows (paths relative to the
reference repo root, Skylarking/MARL).

Rules:
  * Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
    leg may import this package.  The product (``marl_amd``) never does; it fails
    loudly when the HIP library is missing.
  * Parity of this oracle with the true reference is PINNED by the golden fixtures
    under ``tests/golden/`` (generated here by importing the reference itself with
    ``tests/golden/make_golden.py``; see ``tests/test_oracle_golden.py``).
  * Third-party arithmetic used by the reference (torch nn.Linear/GRUCell/ELU/
    sigmoid/bmm, autograd, RMSprop, Adam, clip_grad_norm_) is unpinned upstream;
    the oracle uses torch 2.10 CPU semantics, as the golden vectors do.
"""
