"""HEALPix RING geometry helpers for the Python host mirror (setup-time only; the hot path is in HIP).
Published HEALPix RING scheme; ring ownership as commander3/src/comm_map_mod.f90:193-226."""
import numpy as np


def ring_info(nside, ring):
    """(nphi, z, phi0, startpix) of ring 1..4*nside-1."""
    N = int(nside)
    npix = 12 * N * N
    nr = 4 * N - ring if ring > 2 * N else ring
    if nr < N:
        z = 1.0 - nr * nr / (3.0 * N * N)
        nphi, phi0, start = 4 * nr, np.pi / (4.0 * nr), 2 * nr * (nr - 1)
    else:
        z = 4.0 / 3.0 - 2.0 * nr / (3.0 * N)
        nphi = 4 * N
        phi0 = 0.0 if ((nr - N) & 1) else np.pi / (4.0 * N)
        start = 2 * N * (N - 1) + 4 * N * (nr - N)
    if ring != nr:
        z = -z
        start = npix - start - nphi
    return nphi, z, phi0, start


def pix_z(nside):
    """cos(theta) of every RING pixel."""
    z = np.empty(12 * nside * nside)
    for ring in range(1, 4 * nside):
        nphi, zz, _, start = ring_info(nside, ring)
        z[start:start + nphi] = zz
    return z


def pix_phi(nside):
    """Azimuth of every RING pixel."""
    phi = np.empty(12 * nside * nside)
    for ring in range(1, 4 * nside):
        nphi, _, phi0, start = ring_info(nside, ring)
        phi[start:start + nphi] = phi0 + 2.0 * np.pi * np.arange(nphi) / nphi
    return phi


def rank_rings(nside, rank, nranks):
    """Northern ring numbers owned by ``rank`` (comm_map_mod.f90:197: i = 1+myid, 2*nside, nprocs)."""
    return np.arange(1 + rank, 2 * nside + 1, nranks, dtype=np.int32)


def local_pixels(nside, rings):
    """Full-sky RING indices of the local map of a rank owning ``rings`` (+ mirrors), ascending (:226 sort)."""
    allr = sorted([int(i) for i in rings] + [4 * nside - int(i) for i in rings if i < 2 * nside])
    out = []
    for i in allr:
        nphi, _, _, start = ring_info(nside, i)
        out.append(np.arange(start, start + nphi))
    return np.concatenate(out)
