"""Python host mirror of Commander3's CR module on top of the ``cmdr_*`` C ABI.

Names and argument meaning follow ``commander3/src/comm_cr_mod.f90`` (public list at :36):
``cr_matmulA``, ``cr_invM``, ``cr_computeRHS``, ``solve_cr_eqn_by_CG``; component / band registration mirrors
what ``initialize_signal_mod`` / ``initialize_data_mod`` hand to the solver.  Everything numeric runs in
libcmdr_hip.so; this file only marshals arrays.  No CPU fallback: a missing library or GPU raises ``CmdrError``.
"""
import ctypes

import numpy as np

from . import lib as _libmod
from .lib import CmdrError, check

_dp = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p
ALLREDUCE_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)
ALLREDUCE_STREAM_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

CRIT = {"residual": 0, "fixed_iter": 1, "chisq": 2}  # cpar%cg_conv_crit (comm_cr_mod.f90:220-229)


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


class DeviceArray:
    """fp64 buffer in HBM allocated through ``cmdr_dev_alloc`` (so callers can keep vectors resident)."""

    def __init__(self, L, n, src=None):
        self.L, self.n = L, int(n)
        p = _vp()
        check(L.cmdr_dev_alloc(max(self.n, 1) * 8, ctypes.byref(p)), L)
        self.ptr = p
        if src is not None:
            self.upload(src)

    def upload(self, src):
        a = np.ascontiguousarray(src, dtype=np.float64).reshape(-1)
        assert a.size == self.n, (a.size, self.n)
        check(self.L.cmdr_memcpy_h2d(self.ptr, a.ctypes.data_as(_vp), a.nbytes), self.L)

    def download(self):
        out = np.empty(self.n)
        check(self.L.cmdr_memcpy_d2h(out.ctypes.data_as(_vp), self.ptr, out.nbytes), self.L)
        return out

    def free(self):
        if self.ptr:
            self.L.cmdr_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CRContext:
    """One CR linear system on one GPU (``cmdr_ctx``)."""

    def __init__(self, device=0, _lib=None):
        self.L = _lib if _lib is not None else _libmod.lib()
        h = _vp()
        check(self.L.cmdr_ctx_create(int(device), ctypes.byref(h)), self.L)
        self._h = h
        self.nband = 0
        self.ncomp = 0
        self.band_shape = []  # (npix_local, nmaps)
        self.band_meta = []   # (nside, lmax)
        self.ncr = None
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self.L.cmdr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup ------------------------------------------------------------------------------------------
    def set_rings(self, nside, rings):
        r = np.ascontiguousarray(rings, dtype=np.int32)
        check(self.L.cmdr_ctx_set_rings(self._h, int(nside), r.size, r.ctypes.data_as(ctypes.POINTER(ctypes.c_int))),
              self.L)

    def set_allreduce(self, fn):
        """fn(dev_ptr: int, n: int) sums n doubles at device address dev_ptr over all ranks, in place."""
        def _cb(user, ptr, n):
            fn(ptr, n)
        cb = ALLREDUCE_CB(_cb)
        self._keep.append(cb)
        check(self.L.cmdr_ctx_set_allreduce(self._h, ctypes.cast(cb, _vp), None), self.L)

    def set_allreduce_stream(self, fn):
        """fn(dev_ptr: int, n: int, hip_stream: int) ENQUEUES the in-place sum over ranks on the library's stream
        (RCCL through torch.cuda.ExternalStream in bench.py); no host synchronisation per matvec."""
        def _cb(user, ptr, n, stream):
            fn(ptr, n, stream or 0)
        cb = ALLREDUCE_STREAM_CB(_cb)
        self._keep.append(cb)
        check(self.L.cmdr_ctx_set_allreduce_stream(self._h, ctypes.cast(cb, _vp), None), self.L)

    def set_band_sharding(self, rings_fn, ring_replicas):
        """Band x ring-set hybrid: rings_fn(dev_ptr, n) sums over the ranks holding the same bands (None: one rank)."""
        cbp = None
        if rings_fn is not None:
            def _cb(user, ptr, n):
                rings_fn(ptr, n)
            cb = ALLREDUCE_CB(_cb)
            self._keep.append(cb)
            cbp = ctypes.cast(cb, _vp)
        check(self.L.cmdr_ctx_set_band_sharding(self._h, cbp, None, int(ring_replicas)), self.L)

    # ---- RCCL inside the library (no callbacks): see include/cmdr_hip.h ----------------------------------------
    def rccl_unique_id(self):
        """128-byte ncclUniqueId (create on ONE rank, broadcast with the host language's own means)."""
        buf = ctypes.create_string_buffer(128)
        check(self.L.cmdr_rccl_unique_id(buf), self.L)
        return buf.raw

    def init_rccl(self, unique_id, rank, nranks):
        """Collective: every rank passes the same id.  All sums over ranks then run as ncclAllReduce on the library stream."""
        assert len(unique_id) == 128
        check(self.L.cmdr_ctx_init_rccl(self._h, ctypes.c_char_p(bytes(unique_id)), int(rank), int(nranks)), self.L)

    def rccl_split_rings(self, band_group, ring_index, ring_replicas):
        """Band x ring-set hybrid: communicator of the ranks holding the same bands (ncclCommSplit); collective."""
        check(self.L.cmdr_ctx_rccl_split_rings(self._h, int(band_group), int(ring_index), int(ring_replicas)), self.L)

    def set_vector_slicing(self, rank, nranks):
        """m-sliced CG vectors inside ``solve_cr_eqn_by_CG`` (include/cmdr_hip.h: cmdr_ctx_set_vector_slicing)."""
        check(self.L.cmdr_ctx_set_vector_slicing(self._h, int(rank), int(nranks)), self.L)

    def drop_rccl(self):
        """Destroy the native communicators; the callback forms apply again."""
        check(self.L.cmdr_ctx_drop_rccl(self._h), self.L)

    def rccl_size(self):
        """ncclCommCount read back from the communicator (0: none)."""
        return int(self.L.cmdr_ctx_rccl_size(self._h))

    def set_literal_quirks(self, flag):
        """Reproduce cr_matmulA's stale pmap%alm above a component's lmax (comm_cr_mod.f90:846-861) literally."""
        check(self.L.cmdr_ctx_set_literal_quirks(self._h, int(bool(flag))), self.L)

    def set_only_pol(self, flag):
        check(self.L.cmdr_ctx_set_only_pol(self._h, int(bool(flag))), self.L)

    def add_band(self, nside, lmax, siN, b_l, mb_eff=1.0, sg_mask=None, wring=None):
        """``data(i)``: siN (npix_local[, nmaps]) = 1/rms with 0 in masked pixels; b_l (lmax+1[, nmaps])."""
        siN = _f(np.asarray(siN, dtype=np.float64).reshape(np.shape(siN)[0], -1))
        nmaps = siN.shape[1]
        b_l = _f(np.asarray(b_l, dtype=np.float64).reshape(lmax + 1, -1))
        assert b_l.shape[1] == nmaps
        mk = None if sg_mask is None else _f(np.asarray(sg_mask, dtype=np.float64).reshape(siN.shape))
        w = None if wring is None else np.ascontiguousarray(wring, dtype=np.float64)
        idx = check(self.L.cmdr_band_add(self._h, int(nside), int(lmax), nmaps, _p(siN), _p(b_l), float(mb_eff),
                                         None if mk is None else _p(mk), None if w is None else _p(w)), self.L)
        self.nband += 1
        self.band_shape.append((siN.shape[0], nmaps))
        self.band_meta.append((int(nside), int(lmax)))
        return idx

    def add_comp(self, lmax_amp, nmaps, F_mean, sqrtS_mat=None, sqrtInvS_mat=None, S_mat=None, active=True):
        """One diffuse component; ``sqrtS_mat is None`` means cltype == 'none'.  F_mean: (numband[, nmaps])."""
        F = _f(np.asarray(F_mean, dtype=np.float64).reshape(self.nband, -1))
        assert F.shape[1] == nmaps
        if sqrtS_mat is None:
            idx = check(self.L.cmdr_comp_add(self._h, int(lmax_amp), int(nmaps), -1, None, None, None, _p(F),
                                             int(bool(active))), self.L)
        else:
            a, b, c = _f(sqrtS_mat), _f(sqrtInvS_mat), _f(S_mat)
            lmax_cl = a.shape[2] - 1
            assert a.shape == b.shape == c.shape == (nmaps, nmaps, lmax_cl + 1)
            idx = check(self.L.cmdr_comp_add(self._h, int(lmax_amp), int(nmaps), lmax_cl, _p(a), _p(b), _p(c), _p(F),
                                             int(bool(active))), self.L)
        self.ncomp += 1
        return idx

    def finalize(self):
        check(self.L.cmdr_finalize(self._h), self.L)
        self.ncr = int(self.L.cmdr_ncr(self._h))

    def band_npix(self, b):
        return int(self.L.cmdr_band_npix(self._h, b))

    # ---- preconditioner (initPrecond / update_precond: comm_signal_mod.f90:179, comm_cr_mod.f90:76) -------
    def initPrecond(self, precond="diagonal"):
        """``cpar%cg_precond``: 'diagonal' (initDiffPrecond_diagonal) or 'pseudoinv' (alpha_nu, comm_N_rms_mod.f90:217)."""
        self.precond = precond
        if precond == "diagonal":
            check(self.L.cmdr_precond_init_diag(self._h), self.L)
        elif precond == "pseudoinv":
            check(self.L.cmdr_precond_init_pseudoinv(self._h), self.L)
        else:
            raise ValueError(precond)

    def update_precond(self):
        if getattr(self, "precond", "diagonal") == "pseudoinv":
            check(self.L.cmdr_precond_update_pseudoinv(self._h), self.L)
        else:
            check(self.L.cmdr_precond_update_diag(self._h), self.L)

    def set_lowl_precond(self, comp, lmax_pre_lowl, nside_lowres=None, siN_lowres=None):
        """``CG_LMAX_PRECOND``: low-l dense preconditioner block of a diffuse component (updateLowlPrecond /
        applyLowlPrecond); siN_lowres[b] = ``data(b)%N%siN_lowres`` (full-sky RING, temperature).  Rebuilt by every
        ``update_precond``.  lmax_pre_lowl < 0 removes it."""
        if lmax_pre_lowl < 0:
            check(self.L.cmdr_precond_set_lowl(self._h, int(comp), -1, None, None), self.L)
            return
        ns = (ctypes.c_int * self.nband)(*[int(v) for v in nside_lowres])
        keep = [np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(12 * int(v) ** 2, -1)[:, 0])
                for m, v in zip(siN_lowres, nside_lowres)]
        arr = (_dp * self.nband)(*[_p(a) for a in keep])
        check(self.L.cmdr_precond_set_lowl(self._h, int(comp), int(lmax_pre_lowl), ns, arr), self.L)

    def alpha_nu(self, band):
        out = np.zeros(self.band_shape[band][1])
        check(self.L.cmdr_get_alpha_nu(self._h, int(band), _p(out)), self.L)
        return out

    def set_band_qucov(self, band, iN, siN_mat):
        """comm_N_QUcov: dense inverse covariance and its symmetric square root on the stacked (Q; U) pixels."""
        a = np.ascontiguousarray(iN, dtype=np.float64)
        b = np.ascontiguousarray(siN_mat, dtype=np.float64)
        check(self.L.cmdr_band_set_qucov(self._h, int(band), _p(a), _p(b)), self.L)

    def add_compact(self, nparam, sigma, mean, P, active=True):
        """Compact block (templates / point sources): P = {band: scipy.sparse or dense (ncell_b, nparam)} with
        cell = pix_local + npix_local * stokes.  Call in compList order relative to add_comp."""
        import scipy.sparse as sp
        sg = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float64), (nparam,)))
        mn = np.ascontiguousarray(np.broadcast_to(np.asarray(mean, dtype=np.float64), (nparam,)))
        blk = check(self.L.cmdr_compact_add(self._h, int(nparam), _p(sg), _p(mn), int(bool(active))), self.L)
        for band, M in P.items():
            coo = sp.coo_matrix(M)
            cell = np.ascontiguousarray(coo.row, dtype=np.int64)
            par = np.ascontiguousarray(coo.col, dtype=np.int32)
            val = np.ascontiguousarray(coo.data, dtype=np.float64)
            check(self.L.cmdr_compact_set_band(self._h, blk, int(band), int(val.size),
                                               cell.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                               par.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), _p(val)), self.L)
        return blk

    def set_mixing_map(self, comp, band, F):
        """``F(band,0)%p%map`` of a component with spatially varying mixing (npix_local[, nmaps]); None = F_mean path."""
        if F is None:
            check(self.L.cmdr_comp_set_mixing_map(self._h, int(comp), int(band), None, 0), self.L)
            return
        F = _f(np.asarray(F, dtype=np.float64).reshape(len(F), -1))
        check(self.L.cmdr_comp_set_mixing_map(self._h, int(comp), int(band), _p(F), F.shape[1]), self.L)

    def set_comp_cl(self, comp, sqrtS_mat, sqrtInvS_mat, S_mat):
        """New S tables of a component after ``sampleCls`` -> ``updateS`` (comm_Cl_mod.f90:838-863); arrays
        (nmaps, nmaps, lmax_cl+1).  Follow with ``update_precond`` as the Gibbs loop does."""
        a, b, c = (_f(np.asarray(v, dtype=np.float64)) for v in (sqrtS_mat, sqrtInvS_mat, S_mat))
        check(self.L.cmdr_comp_set_cl(self._h, int(comp), _p(a), _p(b), _p(c)), self.L)

    def set_comp_f_mean(self, comp, F_mean):
        F = _f(np.asarray(F_mean, dtype=np.float64).reshape(len(self.band_shape), -1))
        check(self.L.cmdr_comp_set_f_mean(self._h, int(comp), _p(F)), self.L)

    def set_active(self, comp, active, compact=False):
        """``c%active_samp_group(samp_group)`` of a diffuse component (or compact block) for the next sampling group."""
        f = self.L.cmdr_compact_set_active if compact else self.L.cmdr_comp_set_active
        check(f(self._h, int(comp), int(bool(active))), self.L)

    def set_cl_diag(self, comp, cl):
        cl = _f(np.asarray(cl, dtype=np.float64).reshape(len(cl), -1))
        check(self.L.cmdr_comp_set_cl_diag(self._h, int(comp), _p(cl)), self.L)

    def invN_diag(self, band):
        """``data(band)%N%invN_diag%alm`` (nalm, nmaps) after initPrecond (comm_N_mod.f90:127-197)."""
        lmax, nmaps = self.band_meta[band][1], self.band_shape[band][1]
        out = np.zeros(((lmax + 1) ** 2, nmaps), order="F")
        check(self.L.cmdr_get_invN_diag(self._h, int(band), _p(out)), self.L)
        return out

    # ---- operators on host vectors ---------------------------------------------------------------------------
    def cr_matmulA(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.ncr,)
        y = np.empty(self.ncr)
        check(self.L.cmdr_matmulA(self._h, _p(x), _p(y)), self.L)
        return y

    def cr_invM(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.ncr,)
        y = np.empty(self.ncr)
        check(self.L.cmdr_invM(self._h, _p(x), _p(y)), self.L)
        return y

    def cr_computeRHS(self, operation, resid, xi=None, eta=None, mu=None):
        """resid[i], xi[i]: (npix_local[, nmaps]) per band; eta, mu: (ncr,)."""
        sample = operation == "sample"
        arr = _dp * self.nband
        keep = []

        def cols(lst):
            ptrs = []
            for b, m in enumerate(lst):
                a = _f(np.asarray(m, dtype=np.float64).reshape(self.band_shape[b]))
                keep.append(a)
                ptrs.append(_p(a))
            return arr(*ptrs)
        r = cols(resid)
        x = cols(xi) if sample else None
        e = np.ascontiguousarray(eta, dtype=np.float64) if sample else None
        mm = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        rhs = np.empty(self.ncr)
        check(self.L.cmdr_compute_rhs(self._h, int(sample), r, x, None if e is None else _p(e),
                                      None if mm is None else _p(mm), _p(rhs)), self.L)
        return rhs

    def compute_residual(self, amp, data):
        """``compute_residual(band, cg_samp_group)`` for every band (comm_chisq_mod.f90:196-267): data minus the signal
        of the components OUTSIDE the sampling group (active flag not set), from their amplitudes ``amp`` (ncr, stacked,
        physical units).  Returns the list of residual maps."""
        arr = _dp * self.nband
        keep, din, dout = [], [], []
        for b, m in enumerate(data):
            a = _f(np.asarray(m, dtype=np.float64).reshape(self.band_shape[b]))
            o = np.zeros(self.band_shape[b], order="F")
            keep += [a, o]
            din.append(_p(a)); dout.append(_p(o))
        av = np.ascontiguousarray(amp, dtype=np.float64)
        assert av.shape == (self.ncr,)
        check(self.L.cmdr_compute_residual(self._h, _p(av), arr(*din), arr(*dout)), self.L)
        return keep[1::2]

    MONO_PRIOR = {"monopole": 1, "monopole+dipole": 2}

    def applyMonoDipolePrior(self, comp, amp, nside, mask, prior_type="monopole", b_l_out=None):
        """``applyMonoDipolePrior`` (comm_diffuse_comp_mod.f90:5738-5827; tail of ``sample_amps_by_CG``,
        comm_signal_mod.f90:186-194) on the stacked amplitudes ``amp`` (ncr, physical units): returns (amp_new, mu[4])."""
        a = np.array(amp, dtype=np.float64, order="C")
        assert a.shape == (self.ncr,)
        m = np.ascontiguousarray(mask, dtype=np.float64).ravel()
        bl = None if b_l_out is None else np.ascontiguousarray(b_l_out, dtype=np.float64)
        mu = np.zeros(4)
        check(self.L.cmdr_apply_mono_dipole_prior(self._h, int(comp), _p(a), int(nside), None if bl is None else _p(bl),
                                                  _p(m), m.size, self.MONO_PRIOR[prior_type], _p(mu)), self.L)
        return a, mu

    def applyMonoDipolePrior_dev(self, comp, amp, nside, mask, prior_type="monopole", b_l_out=None):
        """The same on device buffers (``ctx.dev``): ``amp`` is edited in place; returns mu[4]."""
        bl = None if b_l_out is None else np.ascontiguousarray(b_l_out, dtype=np.float64)
        mu = np.zeros(4)
        check(self.L.cmdr_apply_mono_dipole_prior_dev(self._h, int(comp), amp.ptr, int(nside),
                                                      None if bl is None else _p(bl), mask.ptr,
                                                      self.MONO_PRIOR[prior_type], _p(mu)), self.L)
        return mu

    def solve_cr_eqn_by_CG(self, b, conv_crit="fixed_iter", tol=1e-8, miniter=5, maxiter=40, check_freq=1, x0=None):
        """Returns (x, niter, stat, (delta_new, delta0)); x already multiplied by sqrt(S)."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.shape == (self.ncr,)
        x = np.empty(self.ncr)
        x0a = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
        niter, stat = ctypes.c_int(0), ctypes.c_int(0)
        res = np.zeros(2)
        check(self.L.cmdr_solve(self._h, _p(b), _p(x), CRIT[conv_crit], float(tol), int(miniter), int(maxiter),
                                int(check_freq), None if x0a is None else _p(x0a), ctypes.byref(niter), _p(res),
                                ctypes.byref(stat)), self.L)
        return x, niter.value, stat.value, (res[0], res[1])

    # ---- device-resident variants (bench.py) ----------------------------------------------------------------
    def dev(self, n, src=None):
        return DeviceArray(self.L, n, src)

    def host_register(self, a):
        """Page-lock a long-lived, C-contiguous numpy array (cmdr_host_register): the host-pointer entry points then copy
        from / to it by DMA.  Returns the array; ``host_unregister`` before it is freed."""
        assert a.flags["C_CONTIGUOUS"]
        check(self.L.cmdr_host_register(ctypes.c_void_p(a.ctypes.data), a.nbytes), self.L)
        return a

    def host_unregister(self, a):
        check(self.L.cmdr_host_unregister(ctypes.c_void_p(a.ctypes.data)), self.L)

    def cr_matmulA_dev(self, x, y):
        check(self.L.cmdr_matmulA_dev(self._h, x.ptr, y.ptr), self.L)

    def cr_invM_dev(self, x, y):
        check(self.L.cmdr_invM_dev(self._h, x.ptr, y.ptr), self.L)

    def cr_computeRHS_dev(self, operation, resid, xi, eta, mu, rhs):
        sample = operation == "sample"
        arr = _vp * self.nband
        r = arr(*[a.ptr for a in resid])
        x = arr(*[a.ptr for a in xi]) if sample else None
        check(self.L.cmdr_compute_rhs_dev(self._h, int(sample), r, x, eta.ptr if sample else None,
                                          None if mu is None else mu.ptr, rhs.ptr), self.L)

    def solve_dev(self, b, x, conv_crit="fixed_iter", tol=1e-8, miniter=5, maxiter=40, check_freq=1, x0=None):
        niter, stat = ctypes.c_int(0), ctypes.c_int(0)
        res = np.zeros(2)
        check(self.L.cmdr_solve_dev(self._h, b.ptr, x.ptr, CRIT[conv_crit], float(tol), int(miniter), int(maxiter),
                                    int(check_freq), None if x0 is None else x0.ptr, ctypes.byref(niter), _p(res),
                                    ctypes.byref(stat)), self.L)
        return niter.value, stat.value, (res[0], res[1])


def build_context(spec, device=0, rings_by_nside=None, _lib=None):
    """Create a ready CRContext from a problem ``spec`` (see commander_amd.synth.make_problem).
    rings_by_nside: {nside: northern ring numbers} for a ring-sharded rank (maps in spec are then local)."""
    ctx = CRContext(device, _lib=_lib)
    if rings_by_nside:
        for ns, r in rings_by_nside.items():
            ctx.set_rings(ns, r)
    for ib, b in enumerate(spec["bands"]):
        ctx.add_band(b["nside"], b["lmax"], b["siN"], b["b_l"], b.get("mb_eff", 1.0), b.get("sg_mask"), b.get("wring"))
        if b.get("qucov_iN") is not None:
            ctx.set_band_qucov(ib, b["qucov_iN"], b["qucov_siN"])
    kd = 0
    for c in spec["comps"]:      # compList order == stacked-vector order; entries with kind == "compact" are compact blocks
        if c.get("kind") == "compact":
            ctx.add_compact(c["nparam"], c["sigma"], c["mean"], c["P"], c.get("active", True))
            continue
        tabs = (c.get("sqrtS_mat"), c.get("sqrtInvS_mat"), c.get("S_mat"))
        if tabs[0] is not None and (c.get("lmax_prior", -1) >= 0 or c.get("l_apod", 0) != 0):
            # COMP_PRIOR_AMP_LMAX: get_Cl_apod folded into the tables (comm_Cl_mod.f90:572-666)
            tabs = apply_Cl_apod(*tabs, c.get("l_apod", 0), c.get("lmax_prior", -1), _lib=_lib)
        ctx.add_comp(c["lmax"], c["nmaps"], c["F_mean"], tabs[0], tabs[1], tabs[2], c.get("active", True))
        for ib, F in (c.get("F_map") or {}).items():
            ctx.set_mixing_map(kd, ib, F)
        kd += 1
    ctx.finalize()
    return ctx


def getSigmaL(alm, lmax, _lib=None):
    """``comm_map%getSigmaL`` (comm_map_mod.f90:1302-1351): alm (nalm[, nmaps]) -> sigma_l (lmax+1, nspec)."""
    L = _lib if _lib is not None else _libmod.lib()
    a = _f(np.asarray(alm, dtype=np.float64).reshape((lmax + 1) ** 2, -1))
    nmaps = a.shape[1]
    out = np.zeros((lmax + 1, nmaps * (nmaps + 1) // 2), order="F")
    check(L.cmdr_sigma_l(_p(a), int(lmax), nmaps, _p(out)), L)
    return out


def updateS(Dl, lmin, RJ2unit, _lib=None):
    """``comm_Cl%updateS`` (comm_Cl_mod.f90:316-384): Dl (lmax+1, nspec) -> (sqrtS_mat, sqrtInvS_mat, S_mat), each
    (nmaps, nmaps, lmax+1), and the number of multipoles that were not positive definite."""
    L = _lib if _lib is not None else _libmod.lib()
    D = _f(np.asarray(Dl, dtype=np.float64).reshape(len(Dl), -1))
    nmaps = {1: 1, 3: 2, 6: 3}[D.shape[1]]
    rj = np.ascontiguousarray(RJ2unit, dtype=np.float64)
    out = [np.zeros((nmaps, nmaps, D.shape[0]), order="F") for _ in range(3)]
    nbad = check(L.cmdr_cl_update_S(D.shape[0] - 1, nmaps, int(lmin), _p(D), _p(rj), _p(out[0]), _p(out[1]), _p(out[2])), L)
    return out[0], out[1], out[2], nbad


def sampleCls_lookup(Dl, Dl_lookup, lmin_lookup, active, sigma_l, S_mat, RJ2unit, uniform, _lib=None):
    """``sample_Dl_lookup`` (comm_Cl_mod.f90:1063-1145): Dl_lookup (nl, 6, nmodel), active: 6 flags.
    Returns (new Dl, ok, chosen model index)."""
    L = _lib if _lib is not None else _libmod.lib()
    D = _f(np.array(Dl, dtype=np.float64).reshape(len(Dl), 6))
    T = _f(np.asarray(Dl_lookup, dtype=np.float64))
    sg = _f(np.asarray(sigma_l, dtype=np.float64).reshape(D.shape))
    Sm = _f(np.asarray(S_mat, dtype=np.float64).reshape(3, 3, D.shape[0]))
    rj = np.ascontiguousarray(RJ2unit, dtype=np.float64)
    act = (ctypes.c_int * 6)(*[int(bool(a)) for a in active])
    ch = ctypes.c_int(-1)
    rc = check(L.cmdr_cl_sample_lookup(D.shape[0] - 1, int(lmin_lookup), int(lmin_lookup) + T.shape[0] - 1, T.shape[2], _p(T), act,
                                       _p(sg), _p(Sm), _p(rj), float(uniform), _p(D), ctypes.byref(ch)), L)
    return D, rc == 0, ch.value


def apply_Cl_apod(sqrtS_mat, sqrtInvS_mat, S_mat, l_apod, lmax_prior, _lib=None):
    """Fold ``get_Cl_apod`` (comm_Cl_mod.f90:676-704) into copies of the updateS tables: what ``matmulSqrtS`` /
    ``matmulS`` / ``matmulSqrtInvS`` apply per l.  Returns the three scaled tables for ``add_comp`` / ``set_comp_cl``."""
    L = _lib if _lib is not None else _libmod.lib()
    a, b, c = (np.array(v, dtype=np.float64, order="F") for v in (sqrtS_mat, sqrtInvS_mat, S_mat))
    check(L.cmdr_cl_apply_apod(a.shape[2] - 1, a.shape[0], int(l_apod), int(lmax_prior), _p(a), _p(b), _p(c)), L)
    return a, b, c


def sampleCls_binned(Dl, sigma_l, S_mat, RJ2unit, bins, uniforms, _lib=None):
    """``sample_Cls_inverse_wishart2`` for cltype 'binned' (comm_Cl_mod.f90:1008-1249, InvSamp_mod.f90:35-294), no
    lookup branch.  bins: depth-first list of dicts(lmin, lmax, spec (1-based), sample, sigma); one uniform variate per
    sampled bin.  Returns (new Dl, ok, n_uniform_used)."""
    L = _lib if _lib is not None else _libmod.lib()
    D = _f(np.array(Dl, dtype=np.float64).reshape(len(Dl), -1))
    nmaps = {1: 1, 3: 2, 6: 3}[D.shape[1]]
    sg = _f(np.asarray(sigma_l, dtype=np.float64).reshape(D.shape))
    Sm = _f(np.asarray(S_mat, dtype=np.float64).reshape(nmaps, nmaps, D.shape[0]))
    rj = np.ascontiguousarray(RJ2unit, dtype=np.float64)
    arr = (_libmod.ClBin * max(len(bins), 1))()
    for i, b in enumerate(bins):
        arr[i] = _libmod.ClBin(int(b["lmin"]), int(b["lmax"]), int(b["spec"]), int(bool(b["sample"])), float(b["sigma"]))
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    used = ctypes.c_int(0)
    rc = check(L.cmdr_cl_sample_binned(D.shape[0] - 1, nmaps, _p(sg), _p(Sm), _p(rj), len(bins), arr, _p(u), u.size, _p(D),
                                       ctypes.byref(used)), L)
    return D, rc == 0, used.value


def alm_to_chain_order(alm, lmax, _lib=None):
    """Packed a_lm (nalm[, nmaps]) -> the chain file's ``alm`` dataset: float32, index l^2 + l + m (comm_map_mod.f90:712-740)."""
    L = _lib if _lib is not None else _libmod.lib()
    a = _f(np.asarray(alm, dtype=np.float64).reshape((lmax + 1) ** 2, -1))
    out = np.zeros(a.shape, dtype=np.float32, order="F")
    check(L.cmdr_alm_to_chain_order(_p(a), int(lmax), a.shape[1], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))), L)
    return out


def alm_from_chain_order(chain32, lmax, _lib=None):
    L = _lib if _lib is not None else _libmod.lib()
    c = np.asfortranarray(np.asarray(chain32, dtype=np.float32).reshape((lmax + 1) ** 2, -1))
    out = np.zeros(c.shape, order="F")
    check(L.cmdr_alm_from_chain_order(c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(lmax), c.shape[1], _p(out)), L)
    return out


def chain_write_comp(chainfile, iteration, label, alm, lmax, unit_scale=None, sigma_l=None, Dl=None, _lib=None):
    """Write one component's sample into the HDF5 chain file the way ``comm_diffuse_comp%dumpFITS`` does
    (``/<iter>/<label>/amp_alm`` float32 in l^2+l+m order, ``amp_lmax``, ``amp_nmaps``, ``sigma_l``, ``Dl``)."""
    L = _lib if _lib is not None else _libmod.lib()
    a = _f(np.asarray(alm, dtype=np.float64).reshape((lmax + 1) ** 2, -1))
    nmaps = a.shape[1]
    us = None if unit_scale is None else np.ascontiguousarray(unit_scale, dtype=np.float64)
    sg = None if sigma_l is None else _f(np.asarray(sigma_l, dtype=np.float64).reshape(lmax + 1, -1))
    dl = None if Dl is None else _f(np.asarray(Dl, dtype=np.float64).reshape(lmax + 1, -1))
    check(L.cmdr_chain_write_comp(str(chainfile).encode(), int(iteration), label.encode(), _p(a), int(lmax), nmaps,
                                  None if us is None else _p(us), None if sg is None else _p(sg),
                                  None if dl is None else _p(dl)), L)


def chain_read_comp(chainfile, iteration, label, lmax, nmaps, unit_scale=None, read_Dl=False, _lib=None):
    """``initDiffuseHDF``: amplitudes (nalm, nmaps) in the units of ``c%x`` (and Dl (lmax+1, nspec) if asked) of one
    stored sample."""
    L = _lib if _lib is not None else _libmod.lib()
    a = np.zeros(((lmax + 1) ** 2, nmaps), order="F")
    us = None if unit_scale is None else np.ascontiguousarray(unit_scale, dtype=np.float64)
    dl = np.zeros((lmax + 1, nmaps * (nmaps + 1) // 2), order="F") if read_Dl else None
    check(L.cmdr_chain_read_comp(str(chainfile).encode(), int(iteration), label.encode(), int(lmax), int(nmaps),
                                 None if us is None else _p(us), _p(a), None if dl is None else _p(dl)), L)
    return (a, dl) if read_Dl else a
