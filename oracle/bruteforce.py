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
