"""C_l Gibbs step after the amplitude solve: updateS and the `binned` conditional sampler (host-side C++ in
libcmdr_hip.so, no GPU needed) against oracle/cl_oracle.py, plus a known-answer check of the sampler against the
analytic inverse-gamma posterior of a single temperature multipole.  Parity of this step is UNPINNED (the reference holds
no fixtures for it and cannot be built here); the analytic check is the independent anchor."""
import numpy as np
import pytest

from helpers import emul_lib


def _libs():
    """The product library (host functions need no GPU) and the host emulation build of the same sources."""
    import commander_amd.lib as L
    return [("product", L.lib()), ("emul", emul_lib())]


def _teb_dl(lmax, rng):
    l = np.arange(lmax + 1, dtype=np.float64)
    tt = 1000.0 / (1.0 + (l / 30.0) ** 2) + 5.0
    ee = 0.02 * tt + 0.3
    bb = 0.01 * tt + 0.05
    te = 0.4 * np.sqrt(tt * ee) * np.cos(l / 7.0)
    Dl = np.zeros((lmax + 1, 6))
    Dl[:, 0], Dl[:, 1], Dl[:, 3], Dl[:, 5] = tt, te, ee, bb
    Dl[:2] = 0.0                             # no monopole / dipole signal (l < lmin)
    return Dl


@pytest.mark.parametrize("nmaps", [1, 3])
def test_updateS_vs_oracle(nmaps):
    from oracle import cl_oracle
    from commander_amd.cr import updateS
    rng = np.random.default_rng(5 + nmaps)
    lmax, lmin = 40, 2
    Dl = _teb_dl(lmax, rng)
    if nmaps == 1:
        Dl = Dl[:, :1].copy()
        Dl[7, 0] = 0.0                       # a multipole without signal: the `ok` branch
    else:
        Dl[9, 5] = 0.0                       # BB switched off at one l
        Dl[11, 2] = 0.3                      # a TB entry
    RJ = np.array([1.0, 2.5, 2.5])[:nmaps]
    ref = cl_oracle.update_S(Dl, lmin, RJ)
    for name, L in _libs():
        a, b, c, nbad = updateS(Dl, lmin, RJ, _lib=L)
        assert nbad == 0
        for got, want, what in zip((a, b, c), ref, ("sqrtS", "sqrtInvS", "S")):
            err = np.abs(got - want).max() / np.abs(want).max()
            assert err < 1e-13, (name, what, err)
        # what updateS is for: sqrtS sqrtS = S, sqrtS sqrtInvS = 1 on the active block
        for l in range(lmin, lmax + 1):
            act = np.abs(np.diag(a[:, :, l])) > 0
            if not act.any():
                continue
            P = (a[:, :, l] @ b[:, :, l])[np.ix_(act, act)]
            assert np.abs(P - np.eye(act.sum())).max() < 1e-12


def test_updateS_not_positive_definite_is_reported():
    from commander_amd.cr import updateS
    Dl = _teb_dl(8, None)
    Dl[5, 1] = 10.0 * np.sqrt(Dl[5, 0] * Dl[5, 3])     # |TE| > sqrt(TT EE)
    for name, L in _libs():
        a, b, c, nbad = updateS(Dl, 2, np.ones(3), _lib=L)
        assert nbad == 1 and a[0, 0, 5] == -1e30        # compute_hermitian_root's marker (math_tools.f90:640-648)


def test_get_Cl_apod_and_table_folding():
    from oracle import cr_oracle
    from commander_amd.cr import apply_Cl_apod
    for name, L in _libs():
        for l_apod in (0, 7, -7):
            for lmax_prior in (-1, 0, 12):
                for pos in (True, False):
                    for l in range(0, 25):
                        got = L.cmdr_cl_apod(l, l_apod, 20, lmax_prior, int(pos))
                        want = cr_oracle.get_Cl_apod(l, l_apod, 20, lmax_prior, pos)
                        assert abs(got - want) <= 1e-15 * max(1.0, abs(want)), (name, l, l_apod, lmax_prior, pos)
        rng = np.random.default_rng(1)
        a, b, c = (rng.standard_normal((3, 3, 21)) for _ in range(3))
        fa, fb, fc = apply_Cl_apod(a, b, c, 0, 12, _lib=L)
        f = np.array([cr_oracle.get_Cl_apod(l, 0, 20, 12, True) for l in range(21)])
        assert np.allclose(fa, a * f, rtol=1e-15) and np.allclose(fc, c * f * f, rtol=1e-15)
        assert np.allclose(fb, b / f, rtol=1e-15) and f[1] < 0.01 and f[12] == 1.0


def _sigma_from_draw(Dl, lmin, RJ, rng):
    """sigma_l of an a_lm drawn from S: what getSigmaL hands the sampler."""
    from oracle import cl_oracle
    sq, _, _ = cl_oracle.update_S(Dl, lmin, RJ)
    lmax, nspec = Dl.shape[0] - 1, Dl.shape[1]
    nmaps = sq.shape[0]
    sig = np.zeros((lmax + 1, nspec))
    for l in range(lmax + 1):
        a = sq[:, :, l] @ rng.standard_normal((nmaps, 2 * l + 1))
        C = a @ a.T / (2 * l + 1)
        for k, (i, j) in enumerate(cl_oracle.spec_pairs(nmaps)):
            sig[l, k] = C[i, j]
    return sig


def test_sampleCls_binned_T_vs_oracle():
    from oracle import cl_oracle
    from commander_amd.cr import sampleCls_binned
    rng = np.random.default_rng(77)
    lmax, lmin = 48, 2
    Dl = _teb_dl(lmax, rng)[:, :1].copy()
    RJ = np.array([1.7])
    sig = _sigma_from_draw(Dl, lmin, RJ, rng)
    _, _, S = cl_oracle.update_S(Dl, lmin, RJ)
    edges = [(2, 2), (3, 3), (4, 5), (6, 9), (10, 19), (20, 33), (34, 48)]
    bins = [dict(lmin=a, lmax=b, spec=1, sample=(a != 6), sigma=0.1 * Dl[a, 0]) for a, b in edges]
    u = rng.uniform(size=len(bins))
    want = Dl.copy()
    ok, used = cl_oracle.sample_cls_binned(want, sig, S, RJ, bins, u)
    assert ok and used == len(bins) - 1
    assert np.all(want[6:10, 0] == Dl[6:10, 0])          # the bin that is not sampled keeps its value
    assert np.all(want[2:6, 0] != Dl[2:6, 0])
    for name, L in _libs():
        got, ok2, used2 = sampleCls_binned(Dl, sig, S, RJ, bins, u, _lib=L)
        assert ok2 and used2 == used
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < 1e-10, (name, err)


def _teb_case(lmax, edges, seed):
    from oracle import cl_oracle
    rng = np.random.default_rng(seed)
    lmin = 2
    Dl = _teb_dl(lmax, rng)
    RJ = np.array([1.0, 1.3, 1.3])
    sig = _sigma_from_draw(Dl, lmin, RJ, rng)
    _, _, S = cl_oracle.update_S(Dl, lmin, RJ)
    bins = []
    for a, b in edges:                                     # a parent TT bin followed by its TE / EE / BB siblings
        for spec in (1, 2, 4, 6):
            bins.append(dict(lmin=a, lmax=b, spec=spec, sample=True, sigma=0.05 * abs(Dl[a, spec - 1]) + 0.01))
    u = rng.uniform(size=len(bins))
    return Dl, sig, S, RJ, bins, u


def test_sampleCls_binned_TEB_vs_oracle():
    from oracle import cl_oracle
    from commander_amd.cr import sampleCls_binned, updateS
    Dl, sig, S, RJ, bins, u = _teb_case(64, [(8, 15), (16, 31), (32, 64)], 78)
    want = Dl.copy()
    ok, used = cl_oracle.sample_cls_binned(want, sig, S, RJ, bins, u)
    assert ok and used == len(bins)
    for name, L in _libs():
        got, ok2, used2 = sampleCls_binned(Dl, sig, S, RJ, bins, u, _lib=L)
        assert ok2 and used2 == used
        err = np.abs(got - want).max() / np.abs(want).max()
        assert err < 1e-9, (name, err)
        assert np.all(got[:, [2, 4]] == 0.0)               # TB, EB never sampled
        assert np.all(got[:8] == Dl[:8])                   # multipoles outside every bin untouched
        _, _, _, nbad = updateS(got, 2, RJ, _lib=L)
        assert nbad == 0                                   # the priors keep the sampled spectrum positive definite


def test_sampleCls_failure_path_matches_oracle():
    """A TE bin with a dozen modes: the likelihood is flat across the whole prior range, the sampler's grid fills up
    (INVSAMP_MAX_NUM_EVALS) and the reference sets ok = .false. after the first (TT) bin.  Same outcome, same partial
    update and the same number of variates consumed in the product."""
    from oracle import cl_oracle
    from commander_amd.cr import sampleCls_binned
    Dl, sig, S, RJ, bins, u = _teb_case(24, [(2, 3), (4, 7)], 78)
    want = Dl.copy()
    ok, used = cl_oracle.sample_cls_binned(want, sig, S, RJ, bins, u)
    assert not ok and used == 1
    for name, L in _libs():
        got, ok2, used2 = sampleCls_binned(Dl, sig, S, RJ, bins, u, _lib=L)
        assert not ok2 and used2 == 1
        assert np.abs(got - want).max() / np.abs(want).max() < 1e-10, name


def test_sampleCls_argument_errors():
    import commander_amd.lib as lib
    from commander_amd.cr import sampleCls_binned
    Dl, sig, S, RJ, bins, u = _teb_case(24, [(8, 15)], 3)
    for name, L in _libs():
        with pytest.raises(lib.CmdrError, match="uniform variates"):
            sampleCls_binned(Dl, sig, S, RJ, bins, u[:0], _lib=L)
        bad = [dict(bins[0], lmax=99)]
        with pytest.raises(lib.CmdrError, match="bad C_l bin"):
            sampleCls_binned(Dl, sig, S, RJ, bad, u, _lib=L)


@pytest.mark.parametrize("l", [2, 10, 40])
def test_sampler_matches_analytic_inverse_gamma(l):
    """Single temperature multipole: P(C | sigma) ~ C^-(2l+1)/2 exp(-(2l+1) sigma / 2C), an inverse-gamma law with
    shape (2l+1)/2 - 1 and scale (2l+1) sigma / 2.  The sampler inverts its CDF numerically; feed it quantiles."""
    from scipy.stats import invgamma
    from commander_amd.cr import sampleCls_binned
    lmax = l
    fac = l * (l + 1) / (2 * np.pi)
    sigma_C = 3.0                                          # sigma_l in C_l units
    Dl = np.full((lmax + 1, 1), 2.5 * fac)
    S = np.zeros((1, 1, lmax + 1)); S[0, 0, :] = 2.5
    sig = np.zeros((lmax + 1, 1)); sig[l, 0] = sigma_C
    law = invgamma(a=(2 * l + 1) / 2.0 - 1.0, scale=(2 * l + 1) * sigma_C / 2.0)
    bins = [dict(lmin=l, lmax=l, spec=1, sample=True, sigma=0.2 * fac)]
    for name, L in _libs():
        for q in (0.1, 0.5, 0.9):
            got, ok, used = sampleCls_binned(Dl, sig, S, np.ones(1), bins, [q], _lib=L)
            assert ok and used == 1
            c = got[l, 0] / fac
            # the sampler truncates at five sigma and integrates a spline on 10^4 points: ~1e-3 in the quantile
            assert abs(law.cdf(c) - q) < 2e-3, (name, l, q, c, law.ppf(q))


def test_sampler_distribution_kolmogorov_smirnov():
    """400 draws of one multipole (l = 30) from seeded uniforms: the empirical law of the product's samples against the
    analytic inverse-gamma posterior (Kolmogorov-Smirnov).  Since the sampler maps each uniform through its numerical
    CDF, the KS distance measures the CDF error directly (plus the finite-sample term of the uniforms themselves)."""
    from scipy.stats import invgamma, kstest
    import commander_amd.lib as L
    from commander_amd.cr import sampleCls_binned
    l, sigma_C = 30, 1.7
    fac = l * (l + 1) / (2 * np.pi)
    Dl = np.full((l + 1, 1), 1.5 * fac)
    S = np.zeros((1, 1, l + 1)); S[0, 0, :] = 1.5
    sig = np.zeros((l + 1, 1)); sig[l, 0] = sigma_C
    law = invgamma(a=(2 * l + 1) / 2.0 - 1.0, scale=(2 * l + 1) * sigma_C / 2.0)
    bins = [dict(lmin=l, lmax=l, spec=1, sample=True, sigma=0.2 * fac)]
    u = np.random.default_rng(99).uniform(size=400)
    lib = L.lib()
    xs = np.array([sampleCls_binned(Dl, sig, S, np.ones(1), bins, [q], _lib=lib)[0][l, 0] / fac for q in u])
    # the sampler's own CDF error: law.cdf(sample) must reproduce the uniform that produced it
    assert np.abs(law.cdf(xs) - u).max() < 3e-3
    assert kstest(xs, law.cdf).pvalue > 0.01


def test_sampleCls_lookup_vs_oracle():
    """The table branch: five tabulated T/E/B spectra scaled around the truth; the model drawn for a given uniform, the
    copied multipoles and the untouched ones must agree with the oracle; a table whose models are all non-positive
    definite gives ok = .false. on both sides."""
    from oracle import cl_oracle
    from commander_amd.cr import sampleCls_lookup
    rng = np.random.default_rng(5)
    lmax, lo, hi = 40, 2, 29
    Dl = _teb_dl(lmax, rng)
    RJ = np.array([1.0, 1.2, 1.2])
    sig = _sigma_from_draw(Dl, 2, RJ, rng)
    _, _, S = cl_oracle.update_S(Dl, 2, RJ)
    scales = np.array([0.8, 0.9, 1.0, 1.1, 1.25])
    tab = np.stack([Dl[lo:hi + 1] * s for s in scales], axis=2)            # (nl, 6, nmodel)
    active = [1, 1, 0, 1, 0, 1]
    picks = set()
    for name, L in _libs():
        for u in (0.02, 0.35, 0.5, 0.77, 0.999):
            want = Dl.copy()
            ok, ch = cl_oracle.sample_cls_lookup(want, tab, lo, active, sig, S, RJ, u)
            got, ok2, ch2 = sampleCls_lookup(Dl, tab, lo, active, sig, S, RJ, u, _lib=L)
            assert ok and ok2 and ch == ch2, (name, u, ch, ch2)
            assert np.array_equal(got, want)
            assert np.array_equal(got[hi + 1:], Dl[hi + 1:]) and np.array_equal(got[:, [2, 4]], Dl[:, [2, 4]])
            picks.add(ch)
        bad = tab.copy()
        bad[:, 1, :] = 100.0 * np.sqrt(bad[:, 0, :] * bad[:, 3, :])       # |TE| far beyond sqrt(TT EE) everywhere
        ok, _ = cl_oracle.sample_cls_lookup(Dl.copy(), bad, lo, active, sig, S, RJ, 0.5)
        _, ok2, _ = sampleCls_lookup(Dl, bad, lo, active, sig, S, RJ, 0.5, _lib=L)
        assert not ok and not ok2
    assert len(picks) >= 2                                                 # the likelihood is not degenerate
