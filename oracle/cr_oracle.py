"""numpy restatement of Commander3's constrained-realization (CR) linear system and PCG.
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Restates, for *diffuse* components with constant mixing (``F_mean`` fast path) and white per-pixel noise:
  * ``cr_matmulA``            commander3/src/comm_cr_mod.f90:771-1024
  * ``cr_computeRHS``         commander3/src/comm_cr_mod.f90:542-769
  * ``cr_invM``               commander3/src/comm_cr_mod.f90:1026-1077  (diagonal preconditioner only)
  * ``solve_cr_eqn_by_CG``    commander3/src/comm_cr_mod.f90:48-406
  * ``cr_insert/extract_comp``commander3/src/comm_cr_utils.f90:40-113
  * ``evalDiffuseBand`` / ``projectDiffuseBand``  commander3/src/comm_diffuse_comp_mod.f90:2027-2167
  * diagonal preconditioner   init :1167-1252, update :1313-1557, apply :2186-2235
  * ``matmulB``               commander3/src/comm_B_bl_mod.f90:108-127
  * ``comm_N_rms`` invN / sqrtInvN / N   commander3/src/comm_N_rms_mod.f90:264-313
  * ``comm_Cl`` updateS / sqrtS / sqrtInvS  commander3/src/comm_Cl_mod.f90:316-384, 550-704
  * ``compute_invN_lm``       commander3/src/comm_N_mod.f90:127-197
  * ``invert_matrix_with_mask`` commander3/src/math_tools.f90:406-456
Single rank (P=1): every ring and every m is local, so the Fortran local arrays are the global ones.
Random numbers are *inputs* (the Fortran driver owns ``planck_rng``; SURVEY.md §7 "RNG").

PARITY UNPINNED vs. the reference binary (see oracle/__init__.py).
"""
import numpy as np

from . import healpix, sht


# ----------------------------------------------------------------------------- small objects
class Band:
    """``data(i)``: commander3/src/comm_data_mod.f90:33-63 (info, N, B)."""

    def __init__(self, nside, lmax, siN, b_l, mb_eff=1.0, sg_mask=None, wring=None):
        siN = np.asarray(siN, dtype=np.float64)
        if siN.ndim == 1:
            siN = siN[:, None]
        self.nside, self.lmax = int(nside), int(lmax)
        self.nmaps = siN.shape[1]
        self.npix = 12 * nside * nside
        assert siN.shape[0] == self.npix
        self.siN = siN                      # 1/rms, 0 in masked pixels (comm_N_rms_mod.f90:179-193)
        b_l = np.asarray(b_l, dtype=np.float64)
        if b_l.ndim == 1:
            b_l = b_l[:, None]
        assert b_l.shape[0] == lmax + 1
        self.b_l = b_l
        self.mb_eff = float(mb_eff)
        self.sg_mask = None if sg_mask is None else np.asarray(sg_mask, dtype=np.float64).reshape(self.npix, -1)
        self.wring = wring
        self.info = healpix.AlmInfo(lmax)
        self.invN_diag = None
        self.nside_lowres, self.siN_lowres = None, None   # comm_N_rms_mod.f90:250-259 (set_lowres)

    def set_lowres(self, nside_lowres, siN_lowres):
        """``N%siN_lowres`` of comm_N_rms (comm_N_rms_mod.f90:250-259): sqrt(udgrade(siN^2)) * (nside/nside_lowres), the
        noise of the coadded low-resolution pixels; produced by the driver (HEALPix udgrade), consumed by InvN_lowres."""
        s = np.asarray(siN_lowres, dtype=np.float64)
        self.nside_lowres = int(nside_lowres)
        self.siN_lowres = s.reshape(12 * self.nside_lowres ** 2, -1)

    def set_qucov(self, iN, siN_mat):
        """comm_N_QUcov (comm_N_QUcov_mod.f90:236-290): dense inverse covariance iN and its symmetric square root on the
        stacked (Q; U) pixels; temperature is zeroed.  ``self.siN`` then plays siN_diag (preconditioner only)."""
        assert self.nmaps == 3
        self.qucov_iN = np.asarray(iN, dtype=np.float64).reshape(2 * self.npix, 2 * self.npix)
        self.qucov_siN = np.asarray(siN_mat, dtype=np.float64).reshape(2 * self.npix, 2 * self.npix)

    def _dense(self, M, m):                   # matmulInvN_1map / matmulSqrtInvN_1map (:320-385)
        v = M @ np.concatenate([m[:, 1], m[:, 2]])
        return np.stack([np.zeros(self.npix), v[: self.npix], v[self.npix:]], axis=1)

    # comm_N_rms_mod.f90:264-273
    def invN(self, m):
        if getattr(self, "qucov_iN", None) is not None:
            return self._dense(self.qucov_iN, m)
        out = self.siN ** 2 * m
        return out * self.sg_mask if self.sg_mask is not None else out

    # comm_N_rms_mod.f90:304-313
    def sqrtInvN(self, m):
        if getattr(self, "qucov_iN", None) is not None:
            return self._dense(self.qucov_siN, m)
        out = self.siN * m
        return out * self.sg_mask if self.sg_mask is not None else out

    # comm_N_rms_mod.f90:288-301
    def N(self, m):
        with np.errstate(divide="ignore", invalid="ignore"):
            out = np.where(self.siN > 0, m / self.siN ** 2, 0.0)
        return out * self.sg_mask if self.sg_mask is not None else out

    # comm_B_bl_mod.f90:108-127 (trans is ignored by the reference: the operator is diagonal)
    def conv(self, alm, info):
        out = np.zeros_like(alm)
        ok = info.l <= self.lmax
        nm = min(self.b_l.shape[1], alm.shape[1])
        lc = np.minimum(info.l, self.lmax)
        out[:, :nm] = np.where(ok[:, None], alm[:, :nm] * self.b_l[lc, :nm] * self.mb_eff, 0.0)
        # columns beyond the beam's nmaps are left untouched for l<=lmax in the reference
        if alm.shape[1] > nm:
            out[:, nm:] = np.where(ok[:, None], alm[:, nm:], 0.0)
        return out

    def compute_invN_diag(self):
        """comm_N_rms_mod.f90:214-219 + comm_N_mod.f90:127-197."""
        out = np.zeros((self.info.nalm, self.nmaps))
        for j in range(self.nmaps):
            a = sht.YtW(self.nside, self.lmax, self.siN[:, j] ** 2, wring=self.wring)  # :134 YtW_scalar
            out[:, j] = sht.invn_diag(self.nside, self.lmax, a[: self.lmax + 1])
        self.invN_diag = out
        return out


def hermitian_root(A, pw):
    """math_tools.f90:606-662 (dsyevd; returns A(1,1)=-1e30 if any eigenvalue <= 0)."""
    W, V = np.linalg.eigh(A)
    if np.any(W <= 0):
        out = A.copy()
        out[0, 0] = -1e30
        return out
    return (V * W ** pw) @ V.T


def get_Cl_apod(l, l_apod, lmax, lmax_prior, positive):
    """comm_Cl_mod.f90:676-704."""
    alpha = np.log(1e3)
    if l_apod > 0:
        if l <= l_apod:
            f = 1.0
        elif l > lmax:
            f = 0.0
        else:
            f = np.exp(-alpha * (l - l_apod) ** 2 / float(lmax - l_apod + 1) ** 2)
    else:
        if l >= abs(l_apod):
            f = 1.0
        elif l == 0 or l > lmax:
            f = 0.0
        else:
            f = np.exp(-alpha * (abs(l_apod) - l) ** 2 / float(abs(l_apod) - 1) ** 2)
    if lmax_prior >= 0 and l < lmax_prior:
        f = f * (0.5 * (np.cos(np.pi * float(max(l, 1) - lmax_prior) / float(lmax_prior)) + 1.0)) ** 2
    if not positive and f != 0.0:
        f = 1.0 / f
    return float(f)


class Cl:
    """``comm_Cl``: S_mat / sqrtS_mat / sqrtInvS_mat from D_l (comm_Cl_mod.f90:316-384)."""

    def __init__(self, lmax, nmaps, Dl, lmin=0, RJ2unit=None, cltype="power_law", l_apod=0, lmax_prior=-1):
        self.lmax, self.nmaps, self.type = lmax, nmaps, cltype
        self.lmin = lmin
        # get_Cl_apod per l (comm_Cl_mod.f90:676-704): l_apod is never assigned in the reference (=> 0 here);
        # lmax_prior = COMP_PRIOR_AMP_LMAX (:134) switches on a cosine roll-off below it
        self.f_apod = np.array([get_Cl_apod(l, l_apod, lmax, lmax_prior, True) for l in range(lmax + 1)])
        self.g_apod = np.array([get_Cl_apod(l, l_apod, lmax, lmax_prior, False) for l in range(lmax + 1)])
        nspec = nmaps * (nmaps + 1) // 2
        Dl = np.asarray(Dl, dtype=np.float64).reshape(lmax + 1, nspec)
        self.Dl = Dl
        RJ = np.ones(nmaps) if RJ2unit is None else np.asarray(RJ2unit, dtype=np.float64)
        self.S_mat = np.zeros((nmaps, nmaps, lmax + 1))
        self.sqrtS_mat = np.zeros((nmaps, nmaps, lmax + 1))
        self.sqrtInvS_mat = np.zeros((nmaps, nmaps, lmax + 1))
        if cltype == "none":
            return
        for l in range(lmax + 1):
            M = np.zeros((nmaps, nmaps))
            ok = np.zeros(nmaps, dtype=bool)
            k = 0
            for i in range(nmaps):
                for j in range(i, nmaps):
                    if l < lmin:
                        v = 0.0
                    elif l == 0:
                        v = Dl[l, k]
                    else:
                        v = Dl[l, k] / (l * (l + 1) / (2.0 * np.pi))
                    v = v / (RJ[i] * RJ[j])
                    M[i, j] = M[j, i] = v
                    if i == j:
                        ok[i] = Dl[l, k] > 0.0
                    k += 1
            for i in range(nmaps):
                if not ok[i]:
                    M[i, :] = 0.0
                    M[:, i] = 0.0
                    M[i, i] = 1.0
            sq = hermitian_root(M, 0.5)
            isq = hermitian_root(M, -0.5)
            for i in range(nmaps):
                if not ok[i]:
                    sq[i, :] = 0.0
                    sq[:, i] = 0.0
                    isq[i, :] = 0.0
                    isq[:, i] = 0.0
            self.sqrtS_mat[:, :, l] = sq
            self.S_mat[:, :, l] = sq @ sq
            self.sqrtInvS_mat[:, :, l] = isq

    def _apply(self, mats, alm, info, diag=False, inverse=False):
        # matmulSqrtS / matmulSqrtInvS, comm_Cl_mod.f90:588-674: f_apod (or its reciprocal) times the per-l matrix
        if self.type == "none":
            return alm.copy()
        out = np.zeros_like(alm)
        ok = info.l <= self.lmax
        lc = np.minimum(info.l, self.lmax)
        fa = (self.g_apod if inverse else self.f_apod)[lc][:, None]
        if diag:
            d = np.sqrt(np.stack([self.S_mat[j, j, :] for j in range(self.nmaps)], axis=1))  # (lmax+1, nmaps)
            out = np.where(ok[:, None], fa * d[lc, : alm.shape[1]] * alm, 0.0)
        else:
            Ml = np.moveaxis(mats, 2, 0)[lc]  # (nalm, nmaps, nmaps)
            out = np.where(ok[:, None], fa * np.einsum("nij,nj->ni", Ml, alm), 0.0)
        return out

    def getCl(self, l, p):
        """comm_Cl_mod.f90:1440-1456 (p: 0-based Stokes index)."""
        n = self.nmaps
        j = p * n - p * (p - 1) // 2          # diagonal spectrum index of Stokes p (0-based)
        v = self.Dl[l, j] if l == 0 else self.Dl[l, j] / (l * (l + 1) / (2.0 * np.pi))
        return v * self.f_apod[l] ** 2

    def sqrtS(self, alm, info, diag=False):
        return self._apply(self.sqrtS_mat, alm, info, diag)

    def sqrtInvS(self, alm, info):
        return self._apply(self.sqrtInvS_mat, alm, info, inverse=True)


class DiffuseComp:
    """``comm_diffuse_comp`` reduced to what the CR system reads (constant mixing)."""

    def __init__(self, lmax_amp, nmaps, cl, F_mean, active=True, mu=None, nside=None, F_map=None):
        self.lmax_amp, self.nmaps, self.Cl = int(lmax_amp), int(nmaps), cl
        self.F_mean = np.asarray(F_mean, dtype=np.float64).reshape(-1, nmaps)  # (numband, nmaps); det=0
        self.F_null = np.all(self.F_mean == 0.0, axis=1)
        # spatially varying mixing (lmax_ind_mix /= 0): F(band,0)%p%map, {band index: (npix, nmaps)}; bands of this
        # component then take the Y . F . YtW branch (comm_diffuse_comp_mod.f90:2082-2084, 2155-2157)
        self.F_map = {} if F_map is None else {int(k): np.asarray(v, dtype=np.float64).reshape(len(v), -1)
                                               for k, v in F_map.items()}
        self.active = active
        self.info = healpix.AlmInfo(lmax_amp)
        self.mu = mu
        self.nside = nside
        self.cltype = cl.type


class CompactBlock:
    """Pixel-space components with scalar amplitudes: ``comm_template_comp`` (one dense column on one band,
    comm_template_comp_mod.f90:210-270) and ``comm_ptsrc_comp`` (one sparse column per source and Stokes parameter on
    every band, comm_ptsrc_comp_mod.f90:336-428) in one form: parameter p contributes  sum_p P_b[cell, p] a_p  to the
    cell = pix + npix * stokes of band b (evalTemplateBand / evalPtsrcBand) and receives  P_b^t map  (project*Band).
    sigma, mean: the Gaussian prior P_cg / P_x (S^1/2 = sigma, comm_cr_mod.f90:817-833)."""

    def __init__(self, nparam, sigma, mean, P, active=True):
        import scipy.sparse as sp
        self.nparam = int(nparam)
        self.sigma = np.broadcast_to(np.asarray(sigma, dtype=np.float64), (self.nparam,)).copy()
        self.mean = np.broadcast_to(np.asarray(mean, dtype=np.float64), (self.nparam,)).copy()
        self.P = {int(b): sp.csr_matrix(m) for b, m in P.items()}   # band index -> (ncell_b, nparam)
        self.active = active
        self.nmaps = 1
        self.cltype = "compact"


class _LowBand:
    """geometry holder for the low-resolution transforms of the low-l preconditioner (unit ring weights)"""

    def __init__(self, nside):
        self.nside, self.wring = int(nside), None


class CRSystem:
    """The stacked linear system: ``ncr`` / ``ind_comp`` (comm_signal_mod.f90:113-125, comm_cr_utils.f90:25-33)."""

    def __init__(self, bands, comps, only_pol=False, literal_quirks=False):
        self.bands, self.comps = list(bands), list(comps)
        self.only_pol = only_pol
        # True: pmap%alm is re-used across the components of a band exactly as comm_cr_mod.f90:846-861 does (set_alm only
        # overwrites what the component has, comm_map_mod.f90:1193-1210); False: zero-filled (intended semantics)
        self.literal_quirks = literal_quirks
        self.ind_comp = []
        pos = 0
        for c in self.comps:
            n = c.nparam if isinstance(c, CompactBlock) else c.info.nalm * c.nmaps
            self.ind_comp.append((pos, n, c.nmaps))
            pos += n
        self.ncr = pos
        self.precond = None

    # comm_cr_utils.f90:92-113
    def extract(self, k, x):
        pos, n, nmaps = self.ind_comp[k]
        return x[pos:pos + n].reshape(nmaps, n // nmaps).T.copy()

    # comm_cr_utils.f90:57-78
    def insert(self, k, add, alm, x):
        pos, n, nmaps = self.ind_comp[k]
        flat = alm.T.reshape(-1)
        if add:
            x[pos:pos + n] += flat
        else:
            x[pos:pos + n] = flat

    # ---- SHT on (n, nmaps) column blocks; polarised (spin-2) columns are a later round
    @staticmethod
    def _Y(band, alm, lmax):
        """exec_sharp_Y (comm_map_mod.f90:437-455): info%pol (nmaps == 3) -> T spin 0 + (Q,U) one spin-2 call."""
        if alm.shape[1] == 3:
            q, u = sht.sht_spin2(sht.JOB_Y, band.nside, lmax, almE=alm[:, 1], almB=alm[:, 2])
            return np.stack([sht.Y(band.nside, lmax, alm[:, 0]), q, u], axis=1)
        return np.stack([sht.Y(band.nside, lmax, alm[:, j]) for j in range(alm.shape[1])], axis=1)

    @staticmethod
    def _Yt(band, m, lmax):
        """exec_sharp_Yt (comm_map_mod.f90:511-529)."""
        if m.shape[1] == 3:
            e, b = sht.sht_spin2(sht.JOB_Yt, band.nside, lmax, mapQ=m[:, 1], mapU=m[:, 2])
            return np.stack([sht.Yt(band.nside, lmax, m[:, 0]), e, b], axis=1)
        return np.stack([sht.Yt(band.nside, lmax, m[:, j]) for j in range(m.shape[1])], axis=1)

    @staticmethod
    def _YtW(band, m, lmax):
        """exec_sharp_YtW (comm_map_mod.f90:531-555): analysis with ring weights W * 4pi/Npix."""
        if m.shape[1] == 3:
            e, b = sht.sht_spin2(sht.JOB_YtW, band.nside, lmax, mapQ=m[:, 1], mapU=m[:, 2], wring=band.wring)
            return np.stack([sht.YtW(band.nside, lmax, m[:, 0], wring=band.wring), e, b], axis=1)
        return np.stack([sht.YtW(band.nside, lmax, m[:, j], wring=band.wring) for j in range(m.shape[1])], axis=1)

    @staticmethod
    def _WY(band, alm, lmax):
        """exec_sharp_WY (comm_map_mod.f90:457-480): adjoint of the analysis."""
        if alm.shape[1] == 3:
            q, u = sht.sht_spin2(sht.JOB_WY, band.nside, lmax, almE=alm[:, 1], almB=alm[:, 2], wring=band.wring)
            return np.stack([sht.WY(band.nside, lmax, alm[:, 0], wring=band.wring), q, u], axis=1)
        return np.stack([sht.WY(band.nside, lmax, alm[:, j], wring=band.wring) for j in range(alm.shape[1])], axis=1)

    def _mix_map(self, c, ib, alm, lmax):
        """Y -> multiply by F(band)%map -> YtW on (nalm(lmax), nmaps) columns (nmaps = 1 or 3)."""
        b = self.bands[ib]
        mp = self._Y(b, alm, lmax)
        mp = mp * c.F_map[ib][:, : alm.shape[1]]
        return self._YtW(b, mp, lmax)

    def _lmax_all(self):
        lm = -1
        for c in self.comps:
            if c.active and not isinstance(c, CompactBlock):
                lm = max(max(lm, c.lmax_amp), 2)  # comm_cr_mod.f90:815
        return lm

    # ------------------------------------------------------------------ evalDiffuseBand / projectDiffuseBand
    def getBand_alm(self, c, ib, amp_in):
        """comm_diffuse_comp_mod.f90:2027-2109 with alm_out=.true., constant mixing (:2077-2080) then beam (:2089)."""
        b = self.bands[ib]
        out = np.zeros((b.info.nalm, b.nmaps))
        if c.F_null[ib]:
            return out
        nmaps = min(b.nmaps, c.nmaps)
        if ib in c.F_map:
            m = self._mix_map(c, ib, amp_in[:, :nmaps], b.lmax)      # :2082-2084
        else:
            m = amp_in[:, :nmaps] * c.F_mean[ib, :nmaps]             # :2077-2080
        m = b.conv(m, b.info)
        out[:, :nmaps] = m
        return out

    def projectBand_alm(self, c, ib, alm_band):
        """comm_diffuse_comp_mod.f90:2112-2167 with alm_in=.true.: beam -> F_mean -> alm_equal to comp lmax."""
        b = self.bands[ib]
        if c.F_null[ib]:
            return np.zeros((c.info.nalm, c.nmaps))
        nmaps = min(c.nmaps, b.nmaps)
        m = b.conv(alm_band[:, :nmaps], b.info)
        if ib in c.F_map:
            m = self._mix_map(c, ib, m, b.lmax)                      # :2155-2157
        else:
            m = m * c.F_mean[ib, :nmaps]
        return healpix.alm_equal(m, b.info, c.info, nmaps_dst=c.nmaps)

    # ------------------------------------------------------------------ compute_residual
    def apply_mono_dipole_prior(self, k, amp, nside, mask, prior_type="monopole", b_l_out=None, pix=None,
                                allreduce=None):
        """``applyMonoDipolePrior`` (comm_diffuse_comp_mod.f90:5738-5827), the tail of ``sample_amps_by_CG``
        (comm_signal_mod.f90:186-194), on component ``k`` of the stacked amplitudes ``amp`` (physical units).
        Returns (amp_new, mu[0:4]).  ``mask`` = mono_prior_map%map(:,1); ``pix`` = this rank's pixel numbers
        (info%pix, default: the full sky) and ``allreduce`` the sum over ranks of :5766-5767 / :5792-5793."""
        c = self.comps[k]
        if prior_type == "none":                                   # :5746-5748
            return amp.copy(), np.zeros(4)
        alm = self.extract(k, amp)
        # map => comm_map(self%x); B_out%conv(trans=.false.); map%Y   (:5754-5756; only column 1 is read afterwards)
        a1 = alm[:, 0].copy()
        if b_l_out is not None:
            a1 *= np.asarray(b_l_out, dtype=np.float64)[c.info.l]  # comm_B_bl_mod.f90:108-127
        full = sht.Y(nside, c.lmax_amp, a1)
        if pix is None:
            pix = np.arange(12 * nside * nside)
        m = full[pix]
        mask = np.asarray(mask, dtype=np.float64).ravel()
        red = allreduce if allreduce is not None else (lambda v: v)
        mu = np.zeros(4)
        if prior_type == "monopole":                               # :5761-5768
            a = red(np.array([np.sum(m * mask)]))[0]
            b = red(np.array([np.sum(mask)]))[0]
            mu[0] = a / b
        elif prior_type == "monopole+dipole":                      # :5775-5794
            theta, phi = healpix.pix_angles(nside)
            v = np.stack([np.ones(pix.size), np.sin(theta[pix]) * np.cos(phi[pix]), np.sin(theta[pix]) * np.sin(phi[pix]),
                          np.cos(theta[pix])], axis=1)              # v(0) = 1, v(1:3) = pix2vec_ring
            use = mask >= 0.5                                       # 'if (mask < 0.5d0) cycle'
            Amat = red(v[use].T @ v[use])
            bmat = red(v[use].T @ m[use])
            mu = np.linalg.solve(Amat, bmat)                        # solve_system_real = dgesv (math_tools.f90:846-883)
        else:
            raise ValueError("Cross-correlation monopole prior not implemented yet")   # :5804-5806 (the reference stops)
        # subtract in harmonic space (:5811-5824): the loop over i with lm(:, i) tests, as four index lookups
        for (l, mm, sgn, j, f) in ((0, 0, -1.0, 0, np.sqrt(4.0 * np.pi)), (1, -1, -1.0, 2, np.sqrt(4.0 * np.pi / 3.0)),
                                   (1, 0, -1.0, 3, np.sqrt(4.0 * np.pi / 3.0)), (1, 1, +1.0, 1, np.sqrt(4.0 * np.pi / 3.0))):
            i = c.info.lm2i(l, mm)
            if i >= 0:
                alm[i, 0] += sgn * mu[j] * f
        out = amp.copy()
        self.insert(k, False, alm, out)
        return out, mu

    def compute_residual(self, data, amp):
        """commander3/src/comm_chisq_mod.f90:196-267 with cg_samp_group given: for every band, the data map minus the
        signal of the components that are NOT active in the sampling group.  amp: stacked amplitudes (c%x, physical
        units); data[i]: (npix, nmaps).  Returns the list of residual maps."""
        out = []
        for ib, b in enumerate(self.bands):
            res_alm = np.zeros((b.info.nalm, b.nmaps))
            ptsrc = np.zeros(b.npix * b.nmaps)
            nonzero = False
            for k, c in enumerate(self.comps):
                if c.active:                                             # :228-230 skip the group's own components
                    continue
                if isinstance(c, CompactBlock):                          # :246-257 pixel-space getBand
                    if ib in c.P:
                        pos, n, _ = self.ind_comp[k]
                        ptsrc += c.P[ib] @ amp[pos:pos + n]
                    continue
                alm = self.extract(k, amp)
                pm = healpix.alm_equal(alm, c.info, b.info, nmaps_dst=b.nmaps)   # getBand: self%x%alm_equal(m)
                res_alm += self.getBand_alm(c, ib, pm)                   # :238-241
                nonzero = True
            mp = self._Y(b, res_alm, b.lmax) if nonzero else np.zeros((b.npix, b.nmaps))   # :260
            d = np.asarray(data[ib], dtype=np.float64).reshape(b.npix, b.nmaps)
            out.append(d - mp - ptsrc.reshape(b.nmaps, b.npix).T)        # :263
        return out

    # ------------------------------------------------------------------ cr_matmulA
    def matmulA(self, x):
        """commander3/src/comm_cr_mod.f90:771-1024."""
        y = np.zeros(self.ncr)
        sqrtS_x = x.copy()
        for k, c in enumerate(self.comps):  # :797-836
            if not c.active:
                continue
            if isinstance(c, CompactBlock):                          # :817-833  pamp * P_cg(2)
                pos, n, _ = self.ind_comp[k]
                sqrtS_x[pos:pos + n] *= c.sigma
            elif c.cltype != "none":
                alm = self.extract(k, sqrtS_x)
                self.insert(k, False, c.Cl.sqrtS(alm, c.info), sqrtS_x)
        lmax = self._lmax_all()
        for ib, b in enumerate(self.bands):  # :843
            map_alm = np.zeros((b.info.nalm, b.nmaps))
            pmap = np.zeros(b.npix * b.nmaps)                        # compact objects, pixel space (:872-882)
            pmap_alm = np.zeros((b.info.nalm, b.nmaps))              # :847 pmap => comm_map(data(i)%info), allocated once per band
            for k, c in enumerate(self.comps):
                if not c.active:
                    continue
                if isinstance(c, CompactBlock):
                    if ib in c.P:
                        pos, n, _ = self.ind_comp[k]
                        pmap += c.P[ib] @ sqrtS_x[pos:pos + n]
                    continue
                alm = self.extract(k, sqrtS_x)
                alm[c.info.l > b.lmax, :] = 0.0                      # :858-860
                if self.literal_quirks:                              # :861 set_alm: only the component's own (l, m), columns
                    j = b.info.lm2i_vec(c.info.l, c.info.m)
                    ok = j >= 0
                    q = min(b.nmaps, c.nmaps)
                    pmap_alm[j[ok], :q] = alm[ok, :q]
                    pm = pmap_alm.copy()
                else:
                    pm = healpix.alm_equal(alm, c.info, b.info, nmaps_dst=b.nmaps)  # zero-filled (intended semantics)
                map_alm += self.getBand_alm(c, ib, pm)               # :865-867
            if lmax > -1:
                info_buff = healpix.AlmInfo(lmax)                    # :888-892
                buff = healpix.alm_equal(map_alm, b.info, info_buff, nmaps_dst=b.nmaps)
                mp = self._Y(b, buff, lmax)
            else:
                mp = np.zeros((b.npix, b.nmaps))
            mp = mp + pmap.reshape(b.nmaps, b.npix).T                # :897  add compact objects
            mp = b.invN(mp)                                          # :905
            if lmax > -1:
                buff = self._Yt(b, mp, lmax)                         # :914-916
                map_alm = healpix.alm_equal(buff, info_buff, b.info, nmaps_dst=b.nmaps)
            for k, c in enumerate(self.comps):                       # :920-948
                if not c.active:
                    continue
                if isinstance(c, CompactBlock):                      # projectPtsrcBand / projectTemplateBand
                    if ib in c.P:
                        pos, n, _ = self.ind_comp[k]
                        y[pos:pos + n] += c.P[ib].T @ mp.T.reshape(-1)
                    continue
                alm = self.projectBand_alm(c, ib, map_alm)
                alm[c.info.l > b.lmax, :] = 0.0                      # :931-933
                self.insert(k, True, alm, y)
        for k, c in enumerate(self.comps):                           # :957-1008
            if not c.active:
                continue
            if isinstance(c, CompactBlock):                          # :985-1003  sqrtS, then the unit prior term
                pos, n, _ = self.ind_comp[k]
                y[pos:pos + n] = y[pos:pos + n] * c.sigma + x[pos:pos + n]
                continue
            if c.cltype != "none":
                alm = self.extract(k, y)
                self.insert(k, False, c.Cl.sqrtS(alm, c.info), y)
                self.insert(k, True, self.extract(k, x), y)
        return y

    # ------------------------------------------------------------------ cr_computeRHS
    def computeRHS(self, residuals, operation="sample", noise_xi=None, eta=None):
        """commander3/src/comm_cr_mod.f90:542-769.

        residuals[i] : (npix, nmaps) output of compute_residual (comm_chisq_mod.f90:196-267) for band i
        noise_xi[i]  : (npix, nmaps) unit Gaussians in the reference's draw order (Stokes outer, pixel inner :602-608)
        eta          : (ncr,) unit Gaussians for the prior term in stacked order (:704-709)
        """
        rhs = np.zeros(self.ncr)
        for ib, b in enumerate(self.bands):
            mp = np.asarray(residuals[ib], dtype=np.float64).reshape(b.npix, b.nmaps).copy()
            if operation == "sample":                                # :600-609
                mp = b.sqrtInvN(mp)
                mp = mp + np.asarray(noise_xi[ib]).reshape(b.npix, b.nmaps)
                mp = b.sqrtInvN(mp)
            else:
                mp = b.invN(mp)                                      # :611
            alm = self._Yt(b, mp, b.lmax)                            # :615
            alm = b.conv(alm, b.info)                                # :616
            for k, c in enumerate(self.comps):
                if not c.active:
                    continue
                if isinstance(c, CompactBlock):                      # :661-680  Tp = projectBand(map) * sqrtS
                    if ib in c.P:
                        pos, n, _ = self.ind_comp[k]
                        rhs[pos:pos + n] += c.sigma * (c.P[ib].T @ mp.T.reshape(-1))
                    continue
                if c.F_null[ib]:
                    Tm = np.zeros((c.info.nalm, c.nmaps))            # :631-632
                else:
                    Tm = healpix.alm_equal(alm, b.info, c.info, nmaps_dst=c.nmaps)  # :634
                    if ib in c.F_map:                                # :640-650 (at the component's lmax)
                        Fm = np.zeros((b.npix, c.nmaps))
                        nm = min(c.nmaps, b.nmaps)
                        Fm[:, :nm] = c.F_map[ib][:, :nm]
                        mp = self._Y(b, Tm, c.lmax_amp) * Fm
                        Tm = self._YtW(b, mp, c.lmax_amp)
                    else:
                        Tm = Tm * c.F_mean[ib, :]                    # :636-639
                Tm = c.Cl.sqrtS(Tm, c.info)                          # :652
                Tm[c.info.l > b.lmax, :] = 0.0                       # :655-657
                self.insert(k, True, Tm, rhs)                        # :659
        for k, c in enumerate(self.comps):                           # :690-728
            if not c.active or c.cltype == "none":
                continue
            if isinstance(c, CompactBlock):                          # :730-764  eta + P(1)/P(2)
                pos, n, _ = self.ind_comp[k]
                if operation == "sample":
                    rhs[pos:pos + n] += np.asarray(eta, dtype=np.float64)[pos:pos + n]
                rhs[pos:pos + n] += c.mean / c.sigma
                continue
            e = np.zeros((c.info.nalm, c.nmaps))
            if operation == "sample":
                e = self.extract(k, np.asarray(eta, dtype=np.float64))
                if self.only_pol:
                    e[:, 0] = 0.0                                    # :705
            if c.mu is not None:
                e = e + c.Cl.sqrtInvS(c.mu, c.info)                  # :712-725
            self.insert(k, True, e, rhs)
        return rhs

    # ------------------------------------------------------------------ diagonal preconditioner
    def _diffuse(self):
        return [c for c in self.comps if not isinstance(c, CompactBlock)]

    def _compact_precond(self):
        """Dense block of A on every compact block, inverted: delta + sigma (sum_b P_b^t N_b^-1 P_b) sigma.  (The
        reference's initPtsrcPrecond / initTemplatePrecond build approximations of this block -- and the template one
        multiplies maps of different bands, comm_template_comp_mod.f90:356-372; the exact block is used here: a
        preconditioner changes the iterates, not the solution.)"""
        out = {}
        for k, c in enumerate(self.comps):
            if not isinstance(c, CompactBlock):
                continue
            M = np.eye(c.nparam)
            if c.active:
                for ib, P in c.P.items():
                    b = self.bands[ib]
                    w = (b.siN ** 2 * (b.sg_mask if b.sg_mask is not None else 1.0)).T.reshape(-1)
                    Pd = P.toarray()
                    M += c.sigma[:, None] * (Pd.T @ (w[:, None] * Pd)) * c.sigma[None, :]
            out[k] = np.linalg.inv(M)
        return out

    def init_precond_diag(self):
        """initDiffPrecond_diagonal: comm_diffuse_comp_mod.f90:1167-1252."""
        self._compact_inv = self._compact_precond()
        self._all_comps = self.comps
        comps = [c for c in self.comps if not isinstance(c, CompactBlock)]
        self._diff_index = [k for k, c in enumerate(self.comps) if not isinstance(c, CompactBlock)]
        npre = len(comps)
        lmax_pre = max(c.lmax_amp for c in comps)                    # :212
        nmaps_pre = max(c.nmaps for c in comps)                      # :214
        info_pre = healpix.AlmInfo(lmax_pre)
        M0 = np.zeros((info_pre.nalm, nmaps_pre, npre, npre))
        for b in self.bands:
            if b.invN_diag is None:
                b.compute_invN_diag()
        for j in range(nmaps_pre):
            for q, b in enumerate(self.bands):
                if j >= b.nmaps:
                    continue
                i2 = b.info.lm2i_vec(info_pre.l, info_pre.m)
                ok = i2 >= 0
                lc = np.minimum(info_pre.l, b.lmax)
                base = np.where(ok, b.invN_diag[np.maximum(i2, 0), j] * b.b_l[lc, min(j, b.b_l.shape[1] - 1)] ** 2, 0.0)
                for k1, p1 in enumerate(comps):
                    if j >= p1.nmaps:
                        continue
                    for k2, p2 in enumerate(comps):
                        if j >= p2.nmaps:
                            continue
                        sel = (info_pre.l <= p1.lmax_amp) & (info_pre.l <= p2.lmax_amp)
                        M0[:, j, k1, k2] += np.where(sel, base * p1.F_mean[q, j] * p2.F_mean[q, j], 0.0)
        self.precond = dict(info=info_pre, nmaps=nmaps_pre, M0=M0, npre=npre)
        return self.precond

    def update_precond_diag(self):
        """updateDiffPrecond_diagonal: comm_diffuse_comp_mod.f90:1313-1557."""
        P = self.precond
        info_pre, nmaps_pre, npre = P["info"], P["nmaps"], P["npre"]
        M0 = P["M0"]
        # comp2ind: only components with mat(k,k) > 0 take part (:1232-1239)
        present = np.stack([M0[:, :, k, k] > 0.0 for k in range(npre)], axis=2)  # (nalm, nmaps, npre)
        M = M0.copy()
        for k1, c in enumerate(self._diffuse()):                          # right- and left-multiply with sqrt(S), diag form
            if c.cltype == "none":
                continue
            lc = np.minimum(info_pre.l, c.Cl.lmax)
            ok = info_pre.l <= c.Cl.lmax
            d = np.zeros((info_pre.nalm, nmaps_pre))
            for j in range(min(nmaps_pre, c.nmaps)):
                d[:, j] = np.where(ok, c.Cl.f_apod[lc] * np.sqrt(c.Cl.S_mat[j, j, lc]), 0.0)   # sqrtS(diag=.true.) :1371,1409
            M[:, :, :, k1] *= d[:, :, None]
            M[:, :, k1, :] *= d[:, :, None]
        if self.only_pol:
            M[:, 0, :, :] = 0.0                                      # :1428-1433
        for k1, c in enumerate(self._diffuse()):                          # add unity :1452-1470
            if c.cltype == "none":
                continue
            sel = (info_pre.l <= c.lmax_amp)[:, None] & present[:, :, k1]
            M[:, :, k1, k1] += np.where(sel, 1.0, 0.0)
        for k1, c in enumerate(self._diffuse()):                          # :1484-1495
            if c.active:
                continue
            M[:, :, k1, :] = 0.0
            M[:, :, :, k1] = 0.0
        # restrict to 'present' sub-block and invert with mask (math_tools.f90:406-456)
        invM = np.zeros_like(M)
        nalm = info_pre.nalm
        for j in range(nmaps_pre):
            for i in range(nalm):
                idx = np.nonzero(present[i, j])[0]
                if idx.size == 0:
                    continue
                sub = M[i, j][np.ix_(idx, idx)]
                if not np.any(sub != 0.0):
                    invM[i, j][np.ix_(idx, idx)] = sub
                    continue
                sub = sub.copy()
                msk = np.ones(idx.size, dtype=bool)
                for t in range(idx.size):
                    if abs(sub[t, t]) <= 0.0:
                        msk[t] = False
                        sub[t, t] = 1.0
                inv = np.linalg.inv(sub)
                for t in range(idx.size):
                    if not msk[t]:
                        inv[t, t] = 0.0
                invM[i, j][np.ix_(idx, idx)] = inv
        P["invM"] = invM
        P["present"] = present
        return invM

    # ------------------------------------------------------------------ pseudo-inverse preconditioner
    def init_precond_pseudoinv(self):
        """alpha_nu per band and Stokes group (comm_N_rms_mod.f90:217-246): tau = Y Yt siN^2,
        alpha = sqrt(sum tau^2 / sum tau); Q and U share one value."""
        for b in self.bands:
            tau = self._Y(b, self._Yt(b, b.siN ** 2, b.lmax), b.lmax)
            al = np.zeros(b.nmaps)
            st, st2 = tau[:, 0].sum(), (tau[:, 0] ** 2).sum()
            al[0] = np.sqrt(st2 / st) if st > 0 else 0.0
            if b.nmaps == 3:
                st, st2 = tau[:, 1:3].sum(), (tau[:, 1:3] ** 2).sum()
                al[1:3] = np.sqrt(st2 / st) if st > 0 else 0.0
            b.alpha_nu = al
        lmax_pre = max(c.lmax_amp for c in self.comps)
        nmaps_pre = max(c.nmaps for c in self.comps)
        self.precond = dict(type="pseudoinv", info=healpix.AlmInfo(lmax_pre), nmaps=nmaps_pre, npre=len(self.comps))
        return self.precond

    def update_precond_pseudoinv(self):
        """updateDiffPrecond_pseudoinv (comm_diffuse_comp_mod.f90:1560-1658): per (l, Stokes) the pseudo-inverse
        (SVD, relative threshold 1e-12: math_tools.f90:234-292) of U = [alpha_nu b_l F_mean sqrt(C_l) ; prior 1]."""
        P = self.precond
        npre, nb = P["npre"], len(self.bands)
        lmax_pre = P["info"].lmax
        pinv = np.zeros((lmax_pre + 1, P["nmaps"], npre, nb + npre))
        for j in range(P["nmaps"]):
            for l in range(lmax_pre + 1):
                mat = np.zeros((nb + npre, npre))
                for q, b in enumerate(self.bands):
                    if l > b.lmax or j >= b.nmaps:
                        continue
                    for k, c in enumerate(self.comps):
                        if l > c.lmax_amp or j >= c.nmaps or not c.active:
                            continue
                        v = b.alpha_nu[j] * b.b_l[l, min(j, b.b_l.shape[1] - 1)] * c.F_mean[q, j]
                        if c.cltype != "none":
                            v *= np.sqrt(c.Cl.getCl(l, j))
                        mat[q, k] = v
                for k, c in enumerate(self.comps):
                    if c.cltype == "none" or l > c.lmax_amp or not c.active:
                        continue
                    mat[nb + k, k] = 1.0
                pinv[l, j] = np.linalg.pinv(mat, rcond=1e-12)
        P["pinv"] = pinv
        P["ind_pre"] = [k for k, c in enumerate(self.comps) if c.active]
        return pinv

    def _invM_pseudoinv(self, x):
        """applyDiffPrecond_pseudoinv (comm_diffuse_comp_mod.f90:2238-2380)."""
        P = self.precond
        info_pre, nmaps_pre, ind = P["info"], P["nmaps"], P["ind_pre"]
        pinv, nb = P["pinv"], len(self.bands)
        res = np.asarray(x, dtype=np.float64).copy()
        if not ind:
            return res
        yv = np.zeros((len(ind), info_pre.nalm, nmaps_pre))
        for i, k in enumerate(ind):
            c = self.comps[k]
            yv[i][info_pre.lm2i_vec(c.info.l, c.info.m), : c.nmaps] = self.extract(k, x)
        z = np.zeros_like(yv)
        for q, b in enumerate(self.bands):
            sel = b.info.l <= info_pre.lmax
            jpre = info_pre.lm2i_vec(b.info.l[sel], b.info.m[sel])
            lb = b.info.l[sel]
            a = np.zeros((b.info.nalm, b.nmaps))
            for i, k in enumerate(ind):                                  # (U^+)^t
                for p in range(b.nmaps):
                    a[sel, p] += pinv[lb, p, k, q] * yv[i][jpre, p]
            mp = self._WY(b, a, b.lmax)                                  # :2295
            mp = b.N(mp)                                                 # :2297
            a = self._YtW(b, mp, b.lmax) * b.alpha_nu[None, :] ** 2      # :2299-2305
            for i, k in enumerate(ind):                                  # U^+
                for p in range(b.nmaps):
                    z[i][jpre, p] += pinv[lb, p, k, q] * a[sel, p]
        for p in range(nmaps_pre):                                       # prior terms :2328-2352
            Pp = pinv[info_pre.l, p][:, ind][:, :, [nb + k for k in ind]]   # (nalm, n, n): M(ind(a), numband+ind(b))
            w2 = np.einsum("nkj,kn->jn", Pp, yv[:, :, p])                # w2(j) = sum_k M(ind k, nb+ind j) w(k)
            z[:, :, p] += np.einsum("njk,kn->jn", Pp, w2)                # w(j)  = sum_k M(ind j, nb+ind k) w2(k)
        for i, k in enumerate(ind):
            c = self.comps[k]
            self.insert(k, False, z[i][info_pre.lm2i_vec(c.info.l, c.info.m), : c.nmaps], res)
        return res

    def invM(self, x):
        """cr_invM (comm_cr_mod.f90:1026-1077) -> applyDiffPrecond_diagonal (comm_diffuse_comp_mod.f90:2186-2235)."""
        P = self.precond
        if P.get("type") == "pseudoinv":
            return self._invM_pseudoinv(x)
        info_pre, nmaps_pre, npre = P["info"], P["nmaps"], P["npre"]
        gidx = [k for k, c in enumerate(self.comps) if not isinstance(c, CompactBlock)]   # diffuse -> global index
        yv = np.zeros((npre, info_pre.nalm, nmaps_pre))
        for k, c in enumerate(self._diffuse()):
            alm = self.extract(gidx[k], x)
            kidx = info_pre.lm2i_vec(c.info.l, c.info.m)
            yv[k][kidx, : c.nmaps] = alm
        out = yv.copy()
        invM, present = P["invM"], P["present"]
        # y(ind) = M y(ind) over the present sub-block; absent entries pass through unchanged (:2211-2217)
        mv = np.einsum("njab,bnj->anj", invM, yv)
        anyp = present.any(axis=2)                                   # n == 0 -> cycle
        for k in range(npre):
            out[k] = np.where(present[:, :, k] & anyp, mv[k], yv[k])
        res = np.zeros(self.ncr)
        for k, c in enumerate(self._diffuse()):
            kidx = info_pre.lm2i_vec(c.info.l, c.info.m)
            self.insert(gidx[k], False, out[k][kidx, : c.nmaps], res)
        for k, Minv in getattr(self, "_compact_inv", {}).items():    # applyPtsrcPrecond / applyTemplatePrecond
            pos, n, _ = self.ind_comp[k]
            res[pos:pos + n] = Minv @ np.asarray(x)[pos:pos + n]
        for k, (L, Minv) in getattr(self, "_lowl", {}).items():      # low-l preconditioner, comm_cr_mod.f90:1058-1073
            self._apply_lowl(k, L, Minv, np.asarray(x), res)
        return res

    # ------------------------------------------------------------------ low-l preconditioner (CG_LMAX_PRECOND >= 0)
    def set_lowl(self, k, lmax_pre_lowl):
        """comm_diffuse_comp_mod.f90:217-225: CMB component with the diagonal preconditioner type."""
        self._lowl_cfg = getattr(self, "_lowl_cfg", {})
        self._lowl_cfg[k] = int(lmax_pre_lowl)

    def update_lowl(self):
        """updateLowlPrecond (comm_diffuse_comp_mod.f90:5098-5251): dense (L+1)^2 block of A on the temperature a_lm with
        l <= L, probed with unit vectors through low-resolution transforms (nside = N%nside_chisq_lowres, lmax = 2L) and
        the coadded noise InvN_lowres; Cholesky-inverted.  Index of (l, m) in the block: l^2 + l + m."""
        self._lowl = {}
        for k, L in getattr(self, "_lowl_cfg", {}).items():
            c = self.comps[k]
            n = (L + 1) ** 2
            info = healpix.AlmInfo(2 * L)                                       # :5117 (nside = 2 plays no role)
            M = np.zeros((n, n))
            for l in range(L + 1):
                for m in range(-l, l + 1):
                    alm = np.zeros((info.nalm, c.nmaps))
                    alm[info.lm2i(l, m), 0] = 1.0                               # :5124-5126
                    alm = c.Cl.sqrtS(alm, info)                                 # :5128
                    tot = np.zeros((info.nalm, c.nmaps))
                    for ib, b in enumerate(self.bands):                         # :5132-5181
                        nm = min(c.nmaps, b.nmaps)
                        a2 = alm[:, :nm] * c.F_mean[ib, :nm][None, :]
                        a2 = b.conv(a2, info)                                   # :5141
                        lowb = _LowBand(b.nside_lowres)
                        mp = self._Y(lowb, a2, 2 * L)                           # :5147
                        mp = mp * b.siN_lowres[:, :nm] ** 2                     # InvN_lowres, comm_N_rms_mod.f90:276-285
                        a2 = b.conv(self._Yt(lowb, mp, 2 * L), info)            # :5159-5163
                        tot[:, 0] += a2[:, 0] * c.F_mean[ib, 0]                 # :5168-5176: temperature column only
                    tot = c.Cl.sqrtS(tot, info)                                 # :5184
                    i = l * l + l + m
                    for lp in range(L + 1):                                     # :5196-5210
                        for mp_ in range(-lp, lp + 1):
                            j = lp * lp + lp + mp_
                            M[i, j] = tot[info.lm2i(lp, mp_), 0] + (1.0 if i == j else 0.0)
            self._lowl[k] = (L, np.linalg.inv(M))                               # invert_matrix(cholesky) :5229

    def _apply_lowl(self, k, L, Minv, x, res):
        """applyLowlPrecond (:5254-5310): temperature a_lm with l <= L of cr_invM's result are replaced by the dense
        inverse applied to the same entries of the INPUT vector (comm_cr_mod.f90:1064-1067)."""
        c = self.comps[k]
        alm_in = self.extract(k, x)
        alm_out = self.extract(k, res)
        ls = [(l, m) for l in range(L + 1) for m in range(-l, l + 1)]
        idx = np.array([c.info.lm2i(l, m) for l, m in ls])
        y = Minv @ alm_in[idx, 0]                                               # y = matmul(invM_lowl, yloc), invM_lowl(i, q) = invM(i, k_q)
        alm_out[idx, 0] = y
        self.insert(k, False, alm_out, res)

    # ------------------------------------------------------------------ solve_cr_eqn_by_CG
    def compute_chisq(self, x, resid):
        """cr_compute_chisq (comm_cr_mod.f90:408-465) + compute_chisq(chisq_fullsky) (comm_chisq_mod.f90:32-118): the
        sampling group's amplitudes are set to S^1/2 x, the residual of every band against the whole sky model goes
        through sqrtInvN (no samp-group mask) and is squared and summed.  resid[i]: compute_residual(i, samp_group)."""
        sx = x.copy()
        for k, c in enumerate(self.comps):
            if not c.active:
                continue
            if isinstance(c, CompactBlock):
                pos, n, _ = self.ind_comp[k]
                sx[pos:pos + n] *= c.sigma
            elif c.cltype != "none":
                self.insert(k, False, c.Cl.sqrtS(self.extract(k, sx), c.info), sx)
        tot = 0.0
        for ib, b in enumerate(self.bands):
            alm = np.zeros((b.info.nalm, b.nmaps))
            pmap = np.zeros(b.npix * b.nmaps)
            for k, c in enumerate(self.comps):
                if not c.active:
                    continue
                if isinstance(c, CompactBlock):
                    if ib in c.P:
                        pos, n, _ = self.ind_comp[k]
                        pmap += c.P[ib] @ sx[pos:pos + n]
                    continue
                pm = healpix.alm_equal(self.extract(k, sx), c.info, b.info, nmaps_dst=b.nmaps)
                alm += self.getBand_alm(c, ib, pm)
            res = np.asarray(resid[ib], dtype=np.float64).reshape(b.npix, b.nmaps) - self._Y(b, alm, b.lmax) \
                - pmap.reshape(b.nmaps, b.npix).T
            tot += float(np.sum((b.siN * res) ** 2))
        return tot

    def solve(self, b, conv_crit="fixed_iter", tol=1e-8, miniter=5, maxiter=40, check_freq=1, x0=None,
              history=None, resid=None):
        """commander3/src/comm_cr_mod.f90:48-406.  Returns (x, niter, stat); x already multiplied by sqrt(S)."""
        if x0 is None:
            x = np.zeros(self.ncr)                                   # :133-134 cg_init_zero
        else:                                                        # :136-173
            x = np.asarray(x0, dtype=np.float64).copy()
            for k, c in enumerate(self.comps):
                if isinstance(c, CompactBlock):
                    if c.active:
                        pos, n, _ = self.ind_comp[k]
                        x[pos:pos + n] /= c.sigma                      # :160-170
                elif c.active and c.cltype != "none":
                    self.insert(k, False, c.Cl.sqrtInvS(self.extract(k, x), c.info), x)
        r = b - self.matmulA(x)                                      # :201
        d = self.invM(r)                                             # :203
        delta_new = float(r @ d)                                     # :206
        delta0 = float(b @ self.invM(b))                             # :208
        lim = tol * delta0                                           # :220-222
        val = 1e2 * lim
        if conv_crit == "chisq":                                     # :223-226
            lim, val = tol, 1.0
            chisq = self.compute_chisq(x, resid)
        i = 1
        stat = 0
        niter = 0
        while i <= maxiter:                                          # :230
            if i % check_freq == 0:                                  # :236-247
                val = delta_new
                if conv_crit == "chisq":                             # :239-242
                    chisq_prev = chisq
                    chisq = self.compute_chisq(x, resid)
                    val = abs((chisq_prev - chisq) / chisq)
                if val < lim and (i >= miniter or delta_new <= 1e-30 * delta0) and conv_crit != "fixed_iter":
                    break
            q = self.matmulA(d)                                      # :253
            alpha = delta_new / float(d @ q)                         # :254
            x = x + alpha * d                                        # :255
            r = r - alpha * q                                        # :261
            s = self.invM(r)                                         # :266
            delta_old = delta_new
            delta_new = float(r @ s)                                 # :270
            beta = delta_new / delta_old
            d = s + beta * d                                         # :272
            if history is not None:
                history.append(delta_new)
            niter = i
            i += 1
        for k, c in enumerate(self.comps):                           # :350-389
            if isinstance(c, CompactBlock):
                if c.active:
                    pos, n, _ = self.ind_comp[k]
                    x[pos:pos + n] *= c.sigma
            elif c.active and c.cltype != "none":
                self.insert(k, False, c.Cl.sqrtS(self.extract(k, x), c.info), x)
        if i >= maxiter and conv_crit != "fixed_iter":               # :392-395 (Fortran: i == maxiter+1 after a full loop)
            stat = 1
        return x, niter, stat


def cg_solve_2x2_kat():
    """The reference's only known-answer test: commander3/todscripts/wmap/cg_solver.py:30-61
    (A=[[3,2],[2,6]], b=[2,-8], M=I => x=[2,-2]), run through the same recurrence as ``CRSystem.solve``."""
    A = np.array([[3.0, 2.0], [2.0, 6.0]])
    b = np.array([2.0, -8.0])
    x = np.zeros(2)
    r = b - A @ x
    d = r.copy()
    delta_new = r @ d
    delta0 = delta_new
    it = 0
    while it < 1000 and delta_new > 1e-12 * delta0:
        q = A @ d
        alpha = delta_new / (d @ q)
        x = x + alpha * d
        r = r - alpha * q
        s = r.copy()
        delta_old = delta_new
        delta_new = r @ s
        d = s + (delta_new / delta_old) * d
        it += 1
    return x, it


def getSigmaL(alm, lmax):
    """comm_map_mod.f90:1302-1351: sigma_l(l, k) = sum_m a_lm^i a_lm^j / (2l+1), pairs (i<=j) in Commander's order."""
    info = healpix.AlmInfo(lmax)
    alm = np.asarray(alm, dtype=np.float64).reshape(info.nalm, -1)
    nmaps = alm.shape[1]
    out = np.zeros((lmax + 1, nmaps * (nmaps + 1) // 2))
    k = 0
    for i in range(nmaps):
        for j in range(i, nmaps):
            np.add.at(out[:, k], info.l, alm[:, i] * alm[:, j])
            k += 1
    return out / (2.0 * np.arange(lmax + 1)[:, None] + 1.0)
