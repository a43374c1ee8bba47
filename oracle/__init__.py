"""CPU oracle for the SingleBranchNet hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a CPU restatement (numpy for the integer / RNG work, plain
PyTorch-CPU fp32/fp64 ops for the floating-point work) of the reference
algorithms named in SURVEY.md §8(a).  Every function cites the reference
file:line it follows.

Who may import it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — and there only as the checker / timed
CPU baseline, never as the thing shipped or measured as the GPU result.  The
product package (``sibrar---single-branch-recommender_amd``) never imports it
and fails loudly when the HIP library is missing.

Pinning: the oracle is checked against golden vectors generated in the build
container from the real reference (imported from /root/reference, see
``tests/golden/make_golden.py``); the fixtures are committed under
``tests/golden/``.  The one exception is the third-party metric package
``rmet`` (unpinned git dependency, absent here): the metric arithmetic follows
the reference's in-repo definition ``eval/metrics.py`` instead and is
"parity unpinned" with respect to ``rmet`` itself.
"""
