"""CPU oracle for the Commander3 constrained-realization CG path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``commander_amd``) never does.

PARITY UNPINNED vs. the reference binary: Commander3 cannot be compiled in this image (needs HEALPix 3.70 F90,
libsharp2, FFTW, gfortran -- SURVEY.md §8c) and ships no tests or golden vectors for this path.  The oracle is a
restatement of the reference's Fortran (each function cites file:line) on top of a restated libsharp2-style SHT;
it is pinned by (a) a brute-force direct spherical-harmonic sum (``oracle.bruteforce``), (b) the reference's
only known-answer test (2x2 PCG, ``commander3/todscripts/wmap/cg_solver.py:54-61``), (c) hand-derived index
tables, and (d) an exact Wigner-3j evaluation of ``compute_invN_lm`` (``oracle.wigner``).
"""
