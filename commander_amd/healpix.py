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


def rank_rings(nside, rank, nranks, scheme="cyclic", block=64):
    """Northern ring numbers owned by ``rank`` (the southern mirrors are implied).

    ``cyclic``: Commander's own dealing (comm_map_mod.f90:197: i = 1+myid, 2*nside, nprocs).
    ``block``:  blocks of ``block`` adjacent ring pairs dealt boustrophedon (0..N-1, N-1..0, ...) -- the GPU-friendly
    ownership: the 64 lanes of a Legendre wave then hold neighbouring latitudes, whose recursions start at nearly the
    same l (cyclic dealing spreads a wave over a quarter of the hemisphere: 78 % instead of 96 % useful lane steps
    at Nside 1024 / 8 ranks).  The back-and-forth order balances pixels exactly in the polar caps (ring length grows
    linearly) and the Legendre work closely.  Any union of ring pairs is a valid shard for the library."""
    if scheme == "cyclic":
        return np.arange(1 + rank, 2 * nside + 1, nranks, dtype=np.int32)
    assert scheme == "block", scheme
    nr, B = 2 * nside, int(block)
    while B > 1 and nr // B < 2 * nranks:
        B //= 2
    out = []
    for b in range((nr + B - 1) // B):
        pos = b % (2 * nranks)
        if (pos if pos < nranks else 2 * nranks - 1 - pos) == rank:
            out.append(np.arange(b * B + 1, min((b + 1) * B, nr) + 1, dtype=np.int32))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int32)


def local_pixels(nside, rings):
    """Full-sky RING indices of the local map of a rank owning ``rings`` (+ mirrors), ascending (:226 sort)."""
    allr = sorted([int(i) for i in rings] + [4 * nside - int(i) for i in rings if i < 2 * nside])
    out = []
    for i in allr:
        nphi, _, _, start = ring_info(nside, i)
        out.append(np.arange(start, start + nphi))
    return np.concatenate(out)


# ---- RING <-> (x, y, face) of the published HEALPix scheme: what HEALPix' udgrade needs (a pixel's parent at a lower
# resolution is (x >> k, y >> k) on the same base face).  Used to build the low-resolution noise maps that Commander's
# comm_N_rms keeps (siN_lowres, comm_N_rms_mod.f90:250-259) for the synthetic test problems; the library receives them
# from the driver like every other per-band data product.
_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4])
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7])


def ring2xyf(nside, pix):
    pix = np.asarray(pix, dtype=np.int64)
    N = int(nside)
    ncap, npix, nl2 = 2 * N * (N - 1), 12 * N * N, 2 * N
    iring, iphi, kshift, nr, face = (np.zeros(pix.shape, dtype=np.int64) for _ in range(5))
    north, south = pix < ncap, pix >= npix - ncap
    belt = ~(north | south)
    p = pix[north]
    ir = (1 + np.floor(np.sqrt(1.0 + 2.0 * p)).astype(np.int64)) >> 1
    ir = np.where(2 * ir * (ir - 1) > p, ir - 1, ir)                    # guard the float square root
    ir = np.where(2 * ir * (ir + 1) <= p, ir + 1, ir)
    iring[north], iphi[north], nr[north] = ir, p + 1 - 2 * ir * (ir - 1), ir
    face[north] = (iphi[north] - 1) // ir
    p = pix[belt] - ncap
    tmp = p // (4 * N)
    iring[belt], iphi[belt], nr[belt] = tmp + N, p - tmp * 4 * N + 1, N
    kshift[belt] = (tmp + N + N) & 1
    ire, irm = tmp + 1, nl2 + 1 - tmp
    ifm, ifp = (iphi[belt] - ire // 2 + N - 1) // N, (iphi[belt] - irm // 2 + N - 1) // N
    face[belt] = np.where(ifp == ifm, ifp | 4, np.where(ifp < ifm, ifp, ifm + 8))
    p = npix - pix[south]
    ir = (1 + np.floor(np.sqrt(2.0 * p - 1.0)).astype(np.int64)) >> 1
    ir = np.where(2 * ir * (ir - 1) >= p, ir - 1, ir)
    ir = np.where(2 * ir * (ir + 1) < p, ir + 1, ir)
    iphi[south], nr[south] = 4 * ir + 1 - (p - 2 * ir * (ir - 1)), ir
    iring[south] = 2 * nl2 - ir
    face[south] = 8 + (iphi[south] - 1) // ir
    irt = iring - _JRLL[face] * N + 1
    ipt = 2 * iphi - _JPLL[face] * nr - kshift - 1
    ipt = np.where(ipt >= nl2, ipt - 8 * N, ipt)
    return (ipt - irt) >> 1, (-ipt - irt) >> 1, face


def xyf2ring(nside, ix, iy, face):
    N = int(nside)
    ncap, npix, nl4 = 2 * N * (N - 1), 12 * N * N, 4 * N
    jr = _JRLL[face] * N - ix - iy - 1
    north, south = jr < N, jr > 3 * N
    nr = np.where(north, jr, np.where(south, nl4 - jr, N))
    n_before = np.where(north, 2 * nr * (nr - 1), np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - N) * nl4))
    kshift = np.where(north | south, 0, (jr - N) & 1)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, jp)
    jp = np.where(jp < 1, jp + nl4, jp)
    return n_before + jp - 1


def udgrade_sum_ring(m, nside_in, nside_out):
    """Sum of a RING map over the children of every pixel of the coarser RING map (nside_out divides nside_in)."""
    assert nside_in % nside_out == 0
    k = int(np.log2(nside_in // nside_out))
    ix, iy, f = ring2xyf(nside_in, np.arange(12 * nside_in * nside_in))
    parent = xyf2ring(nside_out, ix >> k, iy >> k, f)
    return np.bincount(parent, weights=np.asarray(m, dtype=np.float64), minlength=12 * nside_out * nside_out)
