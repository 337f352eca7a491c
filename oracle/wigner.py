"""Exact Wigner-3j symbols (Racah formula in rational arithmetic) and the literal ``compute_invN_lm`` sum.
TEST INFRASTRUCTURE ONLY.

``invn_diag_3j`` follows commander3/src/comm_N_mod.f90:127-197 term by term (the reference calls SLATEC
``DRC3JJ``, commander3/src/drc3jj.f, for the two 3j families); it is O(lmax^4) here and only meant for
lmax <= ~24, where it pins the Gauss-Legendre evaluation in oracle/sht_oracle.c::orc_invn_diag.
"""
from fractions import Fraction
from math import factorial, sqrt

import numpy as np

from . import healpix


def wigner_3j(j1, j2, j3, m1, m2, m3):
    if m1 + m2 + m3 != 0:
        return 0.0
    if j3 < abs(j1 - j2) or j3 > j1 + j2:
        return 0.0
    if abs(m1) > j1 or abs(m2) > j2 or abs(m3) > j3:
        return 0.0
    f = factorial
    delta = Fraction(f(j1 + j2 - j3) * f(j1 - j2 + j3) * f(-j1 + j2 + j3), f(j1 + j2 + j3 + 1))
    pref = delta * f(j1 + m1) * f(j1 - m1) * f(j2 + m2) * f(j2 - m2) * f(j3 + m3) * f(j3 - m3)
    kmin = max(0, j2 - j3 - m1, j1 - j3 + m2)
    kmax = min(j1 + j2 - j3, j1 - m1, j2 + m2)
    s = Fraction(0)
    for k in range(kmin, kmax + 1):
        den = f(k) * f(j1 + j2 - j3 - k) * f(j1 - m1 - k) * f(j2 + m2 - k) * f(j3 - j2 + m1 + k) * f(j3 - j1 - m2 + k)
        s += Fraction((-1) ** k, den)
    sign = (-1) ** (j1 - j2 - m3)
    # sqrt(pref) * s, keeping precision: pref and s are exact rationals
    val = sign * s * Fraction(1)
    return float(val) * sqrt(float(pref)) if pref < 10 ** 300 else float(val * _isqrt_frac(pref))


def _isqrt_frac(q):
    from math import isqrt

    scale = 10 ** 60
    return Fraction(isqrt(q.numerator * scale * scale // q.denominator), scale)


def invn_diag_3j(nside, lmax, al0):
    """comm_N_mod.f90:153-189 (one Stokes column); returns real-packed (lmax+1)^2 array."""
    info = healpix.AlmInfo(lmax)
    npix = 12.0 * nside * nside
    out = np.zeros(info.nalm)
    for m in range(0, lmax + 1):
        for l in range(m, lmax + 1):
            val = 0.0
            for lp in range(0, min(2 * l, lmax) + 1):
                val += al0[lp] * sqrt(2.0 * lp + 1.0) * wigner_3j(l, l, lp, -m, m, 0) * wigner_3j(l, l, lp, 0, 0, 0)
            val *= (2 * l + 1) / sqrt(4.0 * np.pi) * npix / (4.0 * np.pi)
            if m % 2:
                val = -val
            out[info.lm2i(l, m)] = val
            if m > 0:
                out[info.lm2i(l, -m)] = val
    return out
