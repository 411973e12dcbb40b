"""CPU oracle for the Str2Str sampling hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain-PyTorch (CPU, eager, fp32 with the reference's fp64 islands)
restatement of the reference algorithm for the path named in BASELINE.json's north_star:
SE(3) score network (IPA trunk), SO(3)/R^3 score + reverse step, frame->backbone projection
and the forward_backward sampler loop.  Every function cites the reference file:line it
follows.  It exists only so that tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg can CHECK (and time, as a baseline) the HIP path; nothing under
``str2str_amd/`` imports it, and the product fails loudly when its HIP library is missing.

Pinning: the reference ships no golden vectors or known-answer tests for this path
(SURVEY.md §4, §8c).  The oracle is therefore pinned against the reference ITSELF, imported in
the authoring container (``oracle/validate_against_reference.py``), and against the committed
fixtures in ``tests/golden/*.npz`` that ``tests/golden/make_golden.py`` generated from that
import (``tests/test_oracle_golden.py`` runs everywhere, including the GPU box).
"""
