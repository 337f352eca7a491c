"""Brute-force direct spherical-harmonic sums (scipy).  TEST INFRASTRUCTURE ONLY.

Independent of every recursion / FFT in oracle/sht_oracle.c and in the HIP kernels: builds the dense matrix
``B[p, i] = Y^R_i(theta_p, phi_p)`` from ``scipy.special.sph_harm_y`` for Nside <= 16 and defines
    Y = B,   Yt = B^T,   YtW = B^T diag(w_ring 4pi/Npix),   WY = diag(w_ring 4pi/Npix) B
in Commander's real-packed basis (comm_map_mod.f90:1497-1520: complex a_lm = (v(+m) + i v(-m))/sqrt 2), i.e.
    Y^R_{l,0} = Y_l0,   Y^R_{l,+m} = sqrt2 Re Y_lm,   Y^R_{l,-m} = -sqrt2 Im Y_lm .
This is what pins the sign/normalisation conventions of the oracle ("golden vectors").
"""
import numpy as np

from . import healpix


def basis_matrix(nside, lmax, spin=0):
    from scipy.special import sph_harm_y

    assert spin == 0
    theta, phi = healpix.pix_angles(nside)
    info = healpix.AlmInfo(lmax)
    B = np.empty((theta.size, info.nalm))
    for m in range(0, lmax + 1):
        for l in range(m, lmax + 1):
            y = sph_harm_y(l, m, theta, phi)
            if m == 0:
                B[:, info.lm2i(l, 0)] = y.real
            else:
                B[:, info.lm2i(l, m)] = np.sqrt(2.0) * y.real
                B[:, info.lm2i(l, -m)] = -np.sqrt(2.0) * y.imag
    return B


def ring_weight_per_pixel(nside, wring=None):
    npix = 12 * nside * nside
    w = np.empty(npix)
    for ring in range(1, 4 * nside):
        nphi, z, sth, phi0, start = healpix.ring_info(nside, ring)
        nr = 4 * nside - ring if ring > 2 * nside else ring
        w[start:start + nphi] = (1.0 if wring is None else wring[nr - 1]) * 4.0 * np.pi / npix
    return w


# ----------------------------------------------------------------------------- spin-weighted harmonics (s = +-2)
def spin_Y(s, l, m, theta, phi):
    """Goldberg et al. (1967) closed form of the spin-weighted spherical harmonic sY_lm (explicit finite sum; no
    recursion), the definition HEALPix / libsharp use for polarisation:
        sY_lm = (-1)^m sqrt((2l+1)/4pi (l+m)!(l-m)! / ((l+s)!(l-s)!)) sin^{2l}(theta/2)
                * sum_r C(l-s, r) C(l+s, r+s-m) (-1)^{l-r-s} cot^{2r+s-m}(theta/2) * e^{i m phi}
    """
    from math import comb, factorial, sqrt, pi
    theta = np.asarray(theta, dtype=np.float64)
    pref = (-1) ** m * sqrt((2 * l + 1) / (4 * pi) * factorial(l + m) * factorial(l - m)
                            / (factorial(l + s) * factorial(l - s)))
    sh, ch = np.sin(theta / 2.0), np.cos(theta / 2.0)
    acc = np.zeros_like(theta)
    for r in range(0, l - s + 1):
        k = r + s - m
        if k < 0 or k > l + s:
            continue
        # sin^{2l} cot^{2r+s-m} = sin^{2l-2r-s+m} cos^{2r+s-m}
        acc = acc + comb(l - s, r) * comb(l + s, k) * (-1) ** (l - r - s) * sh ** (2 * l - 2 * r - s + m) * ch ** (2 * r + s - m)
    return pref * acc * np.exp(1j * m * np.asarray(phi))


def basis_matrix_spin2(nside, lmax):
    """Dense real matrix B2 with  [Q; U] = B2 @ [E; B]  in Commander's real-packed basis, from
        (Q +- iU)(p) = sum_{l, m=-l..l} a_{+-2,lm} +-2Y_lm(p),   a_{+-2,lm} = -(E_lm +- i B_lm)
    and reality of Q, U (E_{l,-m} = (-1)^m conj E_lm, same for B).  Rows: Q pixels then U pixels; columns: packed E
    then packed B.  l < 2 columns are zero."""
    theta, phi = healpix.pix_angles(nside)
    info = healpix.AlmInfo(lmax)
    npix, na = theta.size, info.nalm
    B2 = np.zeros((2 * npix, 2 * na))

    def field(E, Bc, l, m):
        """Q + iU produced by complex E_lm = E, B_lm = Bc at (l, m>=0) together with their (l,-m) partners."""
        yp, ym = spin_Y(2, l, m, theta, phi), spin_Y(-2, l, m, theta, phi)
        plus = -(E + 1j * Bc) * yp          # (Q + iU) contribution of (l, m)
        minus = -(E - 1j * Bc) * ym         # (Q - iU) contribution of (l, m)
        if m == 0:
            return plus, minus
        ypn, ymn = spin_Y(2, l, -m, theta, phi), spin_Y(-2, l, -m, theta, phi)
        En, Bn = (-1) ** m * np.conj(E), (-1) ** m * np.conj(Bc)
        return plus - (En + 1j * Bn) * ypn, minus - (En - 1j * Bn) * ymn

    r2 = 1.0 / np.sqrt(2.0)
    for m in range(0, lmax + 1):
        for l in range(max(m, 2), lmax + 1):
            for which in (0, 1):  # 0: E, 1: B
                slots = [(info.lm2i(l, m), 1.0)] if m == 0 else [(info.lm2i(l, m), r2), (info.lm2i(l, -m), 1j * r2)]
                for col, coef in slots:
                    E, Bc = (coef, 0.0) if which == 0 else (0.0, coef)
                    p, mi = field(E, Bc, l, m)
                    q = 0.5 * (p + mi)
                    u = -0.5j * (p - mi)
                    assert np.abs(q.imag).max() < 1e-9 and np.abs(u.imag).max() < 1e-9
                    B2[:npix, which * na + col] = q.real
                    B2[npix:, which * na + col] = u.real
    return B2
