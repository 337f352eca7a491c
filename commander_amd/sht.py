"""Python mirror of the SHT entry points of ``comm_map`` (commander3/src/comm_map_mod.f90:437-579) on top of
``cmdr_sht_*`` (include/cmdr_hip.h)."""
import ctypes

import numpy as np

from .lib import check, lib

JOB_YtW, JOB_Y, JOB_Yt, JOB_WY = 0, 1, 2, 3  # commander3/src/sharp.f90:8-14
_dp = ctypes.POINTER(ctypes.c_double)


class ShtPlan:
    """One (nside, lmax, ring subset) transform plan, cf. ``comm_mapinfo`` (comm_map_mod.f90:134-305)."""

    def __init__(self, nside, lmax, rings=None, wring=None, max_maps=1, pol=False):
        self.nside, self.lmax = int(nside), int(lmax)
        h = ctypes.c_void_p()
        rp, nr = None, 0
        if rings is not None:
            self._rings = np.ascontiguousarray(rings, dtype=np.int32)
            rp, nr = self._rings.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), self._rings.size
        wp = None
        if wring is not None:
            self._w = np.ascontiguousarray(wring, dtype=np.float64)
            assert self._w.shape == (2 * nside,)
            wp = self._w.ctypes.data_as(_dp)
        create = lib().cmdr_sht_plan_create_pol if pol else lib().cmdr_sht_plan_create
        check(create(self.nside, self.lmax, nr, rp, wp, int(max_maps), ctypes.byref(h)))
        self.pol = bool(pol)
        self._h = h
        self.nalm = lib().cmdr_sht_nalm(h)
        self.npix = lib().cmdr_sht_npix(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().cmdr_sht_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def execute(self, job, alm=None, map=None):
        """alm: (nalm, nmaps) / map: (npix, nmaps) Fortran-ordered columns (1-D = one column)."""
        synth = job in (JOB_Y, JOB_WY)
        src = alm if synth else map
        src = np.asarray(src, dtype=np.float64)
        one = src.ndim == 1
        cols = [np.ascontiguousarray(src if one else src[:, k]) for k in range(1 if one else src.shape[1])]
        n_in, n_out = (self.nalm, self.npix) if synth else (self.npix, self.nalm)
        for c in cols:
            assert c.shape == (n_in,), (c.shape, n_in)
        outs = [np.zeros(n_out) for _ in cols]
        arr = _dp * len(cols)
        a_in = arr(*[c.ctypes.data_as(_dp) for c in cols])
        a_out = arr(*[o.ctypes.data_as(_dp) for o in outs])
        if synth:
            check(lib().cmdr_sht_execute(self._h, job, len(cols), a_in, a_out))
        else:
            check(lib().cmdr_sht_execute(self._h, job, len(cols), a_out, a_in))
        return outs[0] if one else np.stack(outs, axis=1)

    def Y(self, alm):
        return self.execute(JOB_Y, alm=alm)

    def Yt(self, map):
        return self.execute(JOB_Yt, map=map)

    def YtW(self, map):
        return self.execute(JOB_YtW, map=map)

    def WY(self, alm):
        return self.execute(JOB_WY, alm=alm)

    def execute_spin2(self, job, almE=None, almB=None, mapQ=None, mapU=None):
        """(Q,U) <-> (E,B), Commander's spin-2 call (comm_map_mod.f90:446-449).  Returns (Q,U) or (E,B)."""
        synth = job in (JOB_Y, JOB_WY)
        if synth:
            e, b = np.ascontiguousarray(almE, dtype=np.float64), np.ascontiguousarray(almB, dtype=np.float64)
            q, u = np.zeros(self.npix), np.zeros(self.npix)
        else:
            q, u = np.ascontiguousarray(mapQ, dtype=np.float64), np.ascontiguousarray(mapU, dtype=np.float64)
            e, b = np.zeros(self.nalm), np.zeros(self.nalm)
        assert e.shape == b.shape == (self.nalm,) and q.shape == u.shape == (self.npix,)
        check(lib().cmdr_sht_execute_spin2(self._h, job, e.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                            q.ctypes.data_as(_dp), u.ctypes.data_as(_dp)))
        return (q, u) if synth else (e, b)
