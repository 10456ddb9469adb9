"""ORACLE -- test infrastructure, NOT product code.

A CPU restatement of the reference's (NazirNayal8/RbA) inference hot path used only as the
checker: ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; nothing under ``rba_amd/`` does, and the product raises if its HIP library is
missing rather than falling back to anything here.

Pinning: every function here is checked in ``tests/test_oracle_golden.py`` against
``tests/golden/*.npz``, which were produced in the build container by the reference's own
Python modules imported from /root/reference (``tests/golden/make_golden.py``).  Pieces whose
reference implementation lives in an absent third-party package (Detectron2's ImageList pad,
sem_seg_postprocess, Conv2d/get_norm wrapper) are restated from Detectron2 v0.6 behaviour and
are "parity unpinned" by reference code -- see DESIGN.md section "Oracle".
"""
