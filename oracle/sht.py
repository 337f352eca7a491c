"""ctypes front-end of oracle/sht_oracle.c.  TEST INFRASTRUCTURE ONLY.

Mirrors the SHT entry points of commander3/src/comm_map_mod.f90:437-579 for one scalar column:
``Y`` (alm->map), ``Yt`` (exact transpose), ``YtW`` (analysis with ring weights * 4pi/Npix), ``WY``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

JOB_YtW, JOB_Y, JOB_Yt, JOB_WY = 0, 1, 2, 3  # commander3/src/sharp.f90:8-14


def build(force=False):
    so = os.path.join(_HERE, "_build", "libsht_oracle.so")
    src = os.path.join(_HERE, "sht_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "libsht_oracle.so")
        if not os.path.exists(so):
            so = build()
        L = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        L.orc_sht.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int]
        L.orc_sht.restype = ctypes.c_int
        L.orc_invn_diag.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int]
        L.orc_invn_diag.restype = ctypes.c_int
        L.orc_sht_spin2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int]
        L.orc_sht_spin2.restype = ctypes.c_int
        L.orc_sht_fast.argtypes = L.orc_sht.argtypes
        L.orc_sht_fast.restype = ctypes.c_int
        L.orc_sht_spin2_fast.argtypes = L.orc_sht_spin2.argtypes
        L.orc_sht_spin2_fast.restype = ctypes.c_int
        L.orc_lm2i.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_lm2i.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def _threads(nthreads, nside):
    """OpenMP team size: the GPU boxes expose far more hardware threads than this job's CPU share (16), and a full
    team per small transform costs more in fork/join than the transform itself."""
    if nthreads:
        return int(nthreads)
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = min(n, int(os.environ.get("ORACLE_THREADS", "16")))
    if nside <= 64:
        n = min(n, 4)
    return max(n, 1)


def _fast(fast):
    """Which Legendre stage: the plain per-ring loops (the restatement as first written) or the same arithmetic blocked
    over ring pairs for SIMD (orc_sht_fast; pinned against the plain form and the brute-force goldens in
    tests/test_oracle.py).  Default: fast, ORACLE_FAST=0 selects the plain loops everywhere."""
    if fast is None:
        return os.environ.get("ORACLE_FAST", "1") != "0"
    return bool(fast)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def nalm(lmax):
    return (lmax + 1) ** 2


def npix(nside):
    return 12 * nside * nside


def sht(job, nside, lmax, alm=None, map=None, wring=None, fft_mode=1, use_mlim=True, nthreads=0, fast=None):
    """Run one scalar SHT.  Returns the output array (map for Y/WY, alm for Yt/YtW)."""
    L = lib()
    fn = L.orc_sht_fast if _fast(fast) else L.orc_sht
    wp = None
    if wring is not None:
        wring = np.ascontiguousarray(wring, dtype=np.float64)
        assert wring.shape == (2 * nside,)
        wp = _p(wring)
    if job in (JOB_Y, JOB_WY):
        a = np.ascontiguousarray(alm, dtype=np.float64)
        assert a.shape == (nalm(lmax),), a.shape
        out = np.zeros(npix(nside))
        rc = fn(job, nside, lmax, wp, _p(a), _p(out), int(fft_mode), int(use_mlim), _threads(nthreads, nside))
    else:
        m = np.ascontiguousarray(map, dtype=np.float64)
        assert m.shape == (npix(nside),), m.shape
        out = np.zeros(nalm(lmax))
        rc = fn(job, nside, lmax, wp, _p(out), _p(m), int(fft_mode), int(use_mlim), _threads(nthreads, nside))
    if rc != 0:
        raise RuntimeError("orc_sht failed")
    return out


def Y(nside, lmax, alm, **kw):
    return sht(JOB_Y, nside, lmax, alm=alm, **kw)


def Yt(nside, lmax, map, **kw):
    return sht(JOB_Yt, nside, lmax, map=map, **kw)


def YtW(nside, lmax, map, wring=None, **kw):
    return sht(JOB_YtW, nside, lmax, map=map, wring=wring, **kw)


def WY(nside, lmax, alm, wring=None, **kw):
    return sht(JOB_WY, nside, lmax, alm=alm, wring=wring, **kw)


def invn_diag(nside, lmax, al0, nthreads=0):
    """compute_invN_lm (commander3/src/comm_N_mod.f90:127-197) by exact Gauss-Legendre quadrature."""
    L = lib()
    al0 = np.ascontiguousarray(al0, dtype=np.float64)
    assert al0.shape == (lmax + 1,)
    out = np.zeros(nalm(lmax))
    L.orc_invn_diag(nside, lmax, _p(al0), _p(out), _threads(nthreads, nside))
    return out


def sht_spin2(job, nside, lmax, almE=None, almB=None, mapQ=None, mapU=None, wring=None, fft_mode=1, use_mlim=True,
              nthreads=0, fast=None):
    """One spin-2 transform (Q,U) <-> (E,B), Commander's polarisation call (comm_map_mod.f90:446-449).
    Returns (mapQ, mapU) for Y/WY, (almE, almB) for Yt/YtW."""
    L = lib()
    wp = None
    if wring is not None:
        wring = np.ascontiguousarray(wring, dtype=np.float64)
        wp = _p(wring)
    if job in (JOB_Y, JOB_WY):
        e = np.ascontiguousarray(almE, dtype=np.float64)
        b = np.ascontiguousarray(almB, dtype=np.float64)
        q, u = np.zeros(npix(nside)), np.zeros(npix(nside))
    else:
        q = np.ascontiguousarray(mapQ, dtype=np.float64)
        u = np.ascontiguousarray(mapU, dtype=np.float64)
        e, b = np.zeros(nalm(lmax)), np.zeros(nalm(lmax))
    fn = L.orc_sht_spin2_fast if _fast(fast) else L.orc_sht_spin2
    rc = fn(job, nside, lmax, wp, _p(e), _p(b), _p(q), _p(u), int(fft_mode), int(use_mlim), _threads(nthreads, nside))
    if rc != 0:
        raise RuntimeError("orc_sht_spin2 failed")
    return (q, u) if job in (JOB_Y, JOB_WY) else (e, b)
