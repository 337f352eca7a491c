"""HEALPix RING geometry + Commander's a_lm index maps (numpy).  TEST INFRASTRUCTURE ONLY.

Geometry follows the published HEALPix RING scheme; Commander builds its pixel
ownership with HEALPix ``in_ring`` and sorts it (commander3/src/comm_map_mod.f90:193-226), i.e. for a single
rank the local map *is* the full RING-ordered map.  The a_lm layout restates
commander3/src/comm_map_mod.f90:228-261 (index build), :1213-1262 (lm2i / i2lm).
"""
import numpy as np


def ring_info(nside, ring):
    """(nphi, z, sth, phi0, startpix) of ring 1..4*nside-1 (north to south)."""
    N = int(nside)
    npix = 12 * N * N
    nr = 4 * N - ring if ring > 2 * N else ring
    if nr < N:
        omz = nr * nr / (3.0 * N * N)
        z = 1.0 - omz
        sth = np.sqrt(omz * (2.0 - omz))
        nphi = 4 * nr
        phi0 = np.pi / (4.0 * nr)
        start = 2 * nr * (nr - 1)
    else:
        z = 4.0 / 3.0 - 2.0 * nr / (3.0 * N)
        sth = np.sqrt((1.0 - z) * (1.0 + z))
        nphi = 4 * N
        phi0 = 0.0 if ((nr - N) & 1) else np.pi / (4.0 * N)
        start = 2 * N * (N - 1) + 4 * N * (nr - N)
    if ring != nr:
        z = -z
        start = npix - start - nphi
    return nphi, z, sth, phi0, start


def pix_angles(nside):
    """theta, phi of every RING pixel (float64 arrays of length 12 nside^2)."""
    npix = 12 * nside * nside
    theta = np.empty(npix)
    phi = np.empty(npix)
    for ring in range(1, 4 * nside):
        nphi, z, sth, phi0, start = ring_info(nside, ring)
        theta[start:start + nphi] = np.arctan2(sth, z)
        phi[start:start + nphi] = phi0 + 2.0 * np.pi * np.arange(nphi) / nphi
    return theta, phi


def pix_ring_z(nside):
    """cos(theta) per pixel."""
    npix = 12 * nside * nside
    z = np.empty(npix)
    for ring in range(1, 4 * nside):
        nphi, zz, sth, phi0, start = ring_info(nside, ring)
        z[start:start + nphi] = zz
    return z


# ----------------------------------------------------------------------------- a_lm layout
class AlmInfo:
    """Commander's ``comm_mapinfo`` harmonic half for ``nprocs`` ranks, rank ``myid``.

    comm_map_mod.f90:228-261: rank r owns m = r, r+P, ...; m=0 block holds l=0..lmax (lmax+1 reals); each m>0
    block holds interleaved (+m, -m) for l=m..lmax.  ``lm[:, i] = (l, m)``, ``mind[m]`` = start (or -1).
    """

    _cache = {}

    def __new__(cls, lmax, myid=0, nprocs=1):
        # the tables are immutable: one instance per (lmax, myid, nprocs) (cr_matmulA builds one per band and call)
        key = (int(lmax), int(myid), int(nprocs))
        inst = cls._cache.get(key)
        if inst is None:
            inst = super().__new__(cls)
            inst._build(*key)
            if len(cls._cache) > 64:
                cls._cache.clear()
            cls._cache[key] = inst
        return inst

    def __init__(self, lmax, myid=0, nprocs=1):
        pass

    def _build(self, lmax, myid, nprocs):
        self.key = (int(lmax), int(myid), int(nprocs))
        self.lmax = int(lmax)
        self.ms = list(range(myid, lmax + 1, nprocs))
        self.mind = -np.ones(lmax + 1, dtype=np.int64)
        ls, ms = [], []
        ind = 0
        for m in self.ms:                    # the loop of comm_map_mod.f90:237-258, one numpy block per m
            self.mind[m] = ind
            if m == 0:
                ls.append(np.arange(0, lmax + 1, dtype=np.int64))
                ms.append(np.zeros(lmax + 1, dtype=np.int64))
                ind += lmax + 1
            else:
                n = lmax - m + 1
                ls.append(np.repeat(np.arange(m, lmax + 1, dtype=np.int64), 2))
                ms.append(np.tile(np.array([m, -m], dtype=np.int64), n))
                ind += 2 * n
        self.nalm = ind
        self.l = np.concatenate(ls) if ls else np.zeros(0, dtype=np.int64)
        self.m = np.concatenate(ms) if ms else np.zeros(0, dtype=np.int64)
        self.lm = np.stack([self.l, self.m])

    def lm2i(self, l, m):
        """comm_map_mod.f90:1213-1246."""
        if l > self.lmax or abs(m) > l:
            return -1
        if self.mind[abs(m)] == -1:
            return -1
        if m == 0:
            return int(self.mind[0] + l)
        i = int(self.mind[abs(m)] + 2 * (l - abs(m)))
        return i + 1 if m < 0 else i

    def lm2i_vec(self, l, m):
        l = np.asarray(l)
        m = np.asarray(m)
        am = np.abs(m)
        ok = (l <= self.lmax) & (am <= l)
        amc = np.where(ok, am, 0)
        base = self.mind[amc]
        ok &= base >= 0
        idx = np.where(m == 0, base + l, base + 2 * (l - am) + (m < 0))
        return np.where(ok, idx, -1)


_EQ_CACHE = {}


def alm_equal(src, src_info, dst_info, nmaps_dst=None):
    """comm_map_mod.f90:1148-1165: copy a_lm between two layouts via (l,m) lookup, zero fill."""
    src = np.asarray(src)
    if src.ndim == 1:
        src = src[:, None]
    nd = src.shape[1] if nmaps_dst is None else nmaps_dst
    out = np.zeros((dst_info.nalm, nd))
    key = (src_info.key, dst_info.key)          # the lookup depends on the two layouts only
    hit = _EQ_CACHE.get(key)
    if hit is None:
        j = src_info.lm2i_vec(dst_info.l, dst_info.m)
        hit = (j, j >= 0)
        if len(_EQ_CACHE) > 256:
            _EQ_CACHE.clear()
        _EQ_CACHE[key] = hit
    j, ok = hit
    q = min(src.shape[1], nd)
    out[ok, :q] = src[j[ok], :q]
    return out
