"""CPU restatement of Commander3's C_l Gibbs step for `binned` spectra -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(commander_amd/csrc/cl_sampler.cpp behind cmdr_cl_update_S / cmdr_cl_sample_binned) never does.

PARITY UNPINNED: the reference cannot be built here and holds no golden vectors for this step;
this file follows the reference text line by line with numpy / LAPACK (numpy.linalg) in place of dsyevd / dpotrf.

Follows
  * comm_Cl%updateS                    commander3/src/comm_Cl_mod.f90:316-384
  * compute_hermitian_root             commander3/src/math_tools.f90:606-662
  * sample_Cls_inverse_wishart2        commander3/src/comm_Cl_mod.f90:1008-1249 (sample_Dl_lookup, sample_Dl_bin,
                                       lnL_invWishart)
  * sample_InvSamp                     commander3/src/InvSamp_mod.f90:35-294
  * spline_plain / splint_plain        commander3/src/spline_1D_mod.f90:109-172, locate_dp: locate_mod.f90:69-102
  * invert_matrix(cholesky, ln_det)    commander3/src/math_tools.f90:76-152

Random numbers: the reference draws exactly one rand_uni per sampled bin (InvSamp_mod.f90:258-261; F(1) = 0 and
F(N) = 1, so the rejection loop never fires for eta in [0, 1]).  The HEALPix generator (rngmod) is not part of
/root/reference; both this oracle and the product take the uniform variates from the caller -- the Fortran driver keeps
its planck_rng handle -- exactly as eta / xi enter cr_computeRHS.
"""
import numpy as np

INVSAMP_MAX_NUM_EVALS = 1000
N_SPLINE = 10000
DELTA_LNL = 12.5
TOLERANCE = 1e-2


def spec_pairs(nmaps):
    return [(i, j) for i in range(nmaps) for j in range(i, nmaps)]


def hermitian_root(A, power):
    """math_tools.f90:606-662 without `trunc`: A(1,1) = -1e30 and return when an eigenvalue is <= 0."""
    W, V = np.linalg.eigh(A)
    if np.any(W <= 0.0):
        A = A.copy()
        A[0, 0] = -1e30
        return A
    return (V * W ** power) @ V.T


def update_S(Dl, lmin, RJ2unit):
    """Dl[(lmax+1), nspec] -> sqrtS, sqrtInvS, S, each [nmaps, nmaps, lmax+1] (comm_Cl_mod.f90:316-384)."""
    Dl = np.asarray(Dl, dtype=np.float64)
    lmax = Dl.shape[0] - 1
    nspec = Dl.shape[1]
    nmaps = {1: 1, 3: 2, 6: 3}[nspec]
    RJ = np.asarray(RJ2unit, dtype=np.float64)
    sqrtS = np.zeros((nmaps, nmaps, lmax + 1))
    sqrtInvS = np.zeros_like(sqrtS)
    S = np.zeros_like(sqrtS)
    for l in range(lmax + 1):
        M = np.zeros((nmaps, nmaps))
        ok = np.zeros(nmaps, dtype=bool)
        for k, (i, j) in enumerate(spec_pairs(nmaps)):
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
        for i in range(nmaps):
            if not ok[i]:
                M[i, :] = 0.0
                M[:, i] = 0.0
                M[i, i] = 1.0
        Minv = M.copy()
        R = hermitian_root(M, 0.5)
        for i in range(nmaps):
            if not ok[i]:
                R[i, :] = 0.0
                R[:, i] = 0.0
        sqrtS[:, :, l] = R
        S[:, :, l] = R @ R
        Ri = hermitian_root(Minv, -0.5)
        for i in range(nmaps):
            if not ok[i]:
                Ri[i, :] = 0.0
                Ri[:, i] = 0.0
        sqrtInvS[:, :, l] = Ri
    return sqrtS, sqrtInvS, S


def sigma_l_matrix(sigma_l_vec, nmaps):
    """getSigmaL's sigma_l_mat from the packed form, with the zero-diagonal fix of comm_Cl_mod.f90:1026-1030."""
    lmax = sigma_l_vec.shape[0] - 1
    M = np.zeros((nmaps, nmaps, lmax + 1))
    for k, (i, j) in enumerate(spec_pairs(nmaps)):
        M[i, j, :] = sigma_l_vec[:, k]
        M[j, i, :] = sigma_l_vec[:, k]
    for i in range(nmaps):
        z = M[i, i, :] == 0.0
        M[i, i, z] = 1.0
    return M


def _spline(x, y):
    """spline_plain with yp1 = ypn = 1e30 (natural), tridag = Thomas algorithm."""
    n = len(x)
    a = np.zeros(n); b = np.zeros(n); c = np.zeros(n); r = np.zeros(n)
    c[:n - 1] = x[1:] - x[:-1]
    r[:n - 1] = 6.0 * ((y[1:] - y[:-1]) / c[:n - 1])
    r[1:n - 1] = r[1:n - 1] - r[0:n - 2]
    a[1:n - 1] = c[0:n - 2]
    b[1:n - 1] = 2.0 * (c[1:n - 1] + a[1:n - 1])
    b[0] = b[n - 1] = 1.0
    r[0] = 0.0; c[0] = 0.0
    r[n - 1] = 0.0; a[n - 1] = 0.0
    # tridag(a(2:n), b, c(1:n-1), r) -> u
    u = np.zeros(n)
    gam = np.zeros(n)
    bet = b[0]
    u[0] = r[0] / bet
    for j in range(1, n):
        gam[j] = c[j - 1] / bet
        bet = b[j] - a[j] * gam[j]
        u[j] = (r[j] - a[j] * u[j - 1]) / bet
    for j in range(n - 2, -1, -1):
        u[j] -= gam[j + 1] * u[j + 1]
    return u


def _locate(xx, x):
    n = len(xx)
    if x == xx[0]:
        return 1
    if x == xx[n - 1]:
        return n - 1
    return int(np.searchsorted(xx, x, side="right"))   # number of xx <= x == NR's jl for ascending xx


def _splint(xa, ya, y2a, x):
    n = len(xa)
    klo = max(min(_locate(xa, x), n - 1), 1)
    khi = klo + 1
    h = xa[khi - 1] - xa[klo - 1]
    a = (xa[khi - 1] - x) / h
    b = (x - xa[klo - 1]) / h
    return a * ya[klo - 1] + b * ya[khi - 1] + ((a ** 3 - a) * y2a[klo - 1] + (b ** 3 - b) * y2a[khi - 1]) * (h ** 2) / 6.0


def sample_invsamp(eta, x_in, lnL, prior, tol=TOLERANCE):
    """InvSamp_mod.f90:35-294 (no precomputed grid, no optimize).  Returns (sample, status, n_eval, used_eta)."""
    prior = list(prior)
    xs = list(x_in)
    ys = [lnL(x) for x in xs]
    stat = 0

    def insert(xn, yn):
        nonlocal stat
        if len(xs) == INVSAMP_MAX_NUM_EVALS:
            stat += 1
            return
        i = 0
        while i < len(xs) and not (xn < xs[i]):
            i += 1
        xs.insert(i, xn)
        ys.insert(i, yn)

    guard = 0
    while ys[0] > ys[1] and xs[0] > prior[0] and stat == 0:
        xn = 0.5 * (xs[0] + prior[0])
        insert(xn, lnL(xn))
    while ys[-1] > ys[-2] and xs[-1] < prior[1] and stat == 0:
        xn = min(xs[-1] + 1.61803 * (xs[-1] - xs[-2]), prior[1])
        insert(xn, lnL(xn))
    if stat != 0:
        return 1e30, stat, len(xs), False
    peak = max(ys)
    while peak - ys[0] < DELTA_LNL and xs[0] > prior[0] and stat == 0:
        xn = 0.5 * (xs[0] + prior[0])
        insert(xn, lnL(xn))
    while peak - ys[-1] < DELTA_LNL and xs[-1] < prior[1] and stat == 0:
        xn = min(xs[-1] + 1.61803 * (xs[-1] - xs[-2]), prior[1])
        insert(xn, lnL(xn))
    if stat != 0:
        return 1e30, stat, len(xs), False

    eps = 1e30
    it = 0
    while eps > tol:
        it += 1
        m = len(xs)
        peak = max(ys)
        xsp = np.array(xs); ysp = np.array(ys)
        y2 = _spline(xsp, ysp)
        eps = 0.0
        # the loop runs over the *current* S_n(i-1), S_n(i) (which shift as points are inserted above i) but the
        # frozen x_spline -- InvSamp_mod.f90:166-177
        for i in range(m, 1, -1):
            if peak - ys[i - 2] < DELTA_LNL or peak - ys[i - 1] < DELTA_LNL:
                xn = 0.5 * (xsp[i - 2] + xsp[i - 1])
                yn = lnL(xn)
                ysn = _splint(xsp, ysp, y2, xn)
                eps = max(abs(yn - ysn), eps)
                if abs(yn - ysn) > tol:
                    insert(xn, yn)
            if stat != 0:
                break
        if it > 100:
            raise RuntimeError("InvSamp: no convergence in 100 refinements (the reference stops here)")
        if stat != 0:
            break
    if stat != 0:
        return 1e30, stat, len(xs), False

    xn_ = np.array(xs); yn_ = np.array(ys)
    n = len(xs)
    y2 = _spline(xn_, yn_)
    peak = yn_.max()
    a, b = 1, n
    while peak - yn_[a] > DELTA_LNL and yn_[a] > yn_[a - 1]:
        a += 1
    while peak - yn_[b - 2] > DELTA_LNL and yn_[b - 2] > yn_[b - 1]:
        b -= 1
    x_min, x_max = xn_[a - 1], xn_[b - 1]
    dx = (x_max - x_min) / (N_SPLINE - 1.0)
    x = x_min + dx * np.arange(N_SPLINE, dtype=np.float64)
    P = np.array([_splint(xn_, yn_, y2, xi) for xi in x])
    P = np.exp(P - P.max())
    F = np.zeros(N_SPLINE)
    F[1] = dx * 0.5 * (P[0] + P[1])
    for j in range(2, N_SPLINE):
        F[j] = F[j - 1] + dx * 0.5 * (P[j - 1] + P[j])
    F = F / F[-1]
    i = 2
    while eta > F[i - 1] and i < N_SPLINE:
        i += 1
    if i == N_SPLINE:
        s = x[-1]
    else:
        s = x[i - 2] + (eta - F[i - 2]) * (x[i - 1] - x[i - 2]) / (F[i - 1] - F[i - 2])
    if s != s:
        return 1e30, 1, n, True
    return max(min(s, prior[1]), prior[0]), 0, n, True


def _inv_chol(S):
    """invert_matrix(S, cholesky=.true., status, ln_det): (inverse, status, ln_det)."""
    try:
        L = np.linalg.cholesky(S)
    except np.linalg.LinAlgError:
        return S, 1, -1e30
    ln_det = 2.0 * np.sum(np.log(np.diag(L)))
    Li = np.linalg.inv(L)
    return Li.T @ Li, 0, ln_det


def sample_cls_binned(Dl, sigma_l_vec, S_mat, RJ2unit, bins, uniforms):
    """sample_Cls_inverse_wishart2 without the lookup branch (comm_Cl_mod.f90:1008-1249).

    Dl[(lmax+1), nspec] is updated in place; bins = depth-first list of dicts(lmin, lmax, spec(1-based), sample, sigma)
    -- the order sample_Dl_bin visits the bins2 tree; one uniform variate is consumed per sampled bin.
    Returns (ok, n_uniform_used)."""
    nspec = Dl.shape[1]
    nmaps = {1: 1, 3: 2, 6: 3}[nspec]
    pairs = spec_pairs(nmaps)
    sig = sigma_l_matrix(np.asarray(sigma_l_vec), nmaps)
    RJ = np.asarray(RJ2unit, dtype=np.float64)
    used = 0
    for bn in bins:
        if not bn["sample"]:
            continue
        lo, hi, spec = bn["lmin"], bn["lmax"], bn["spec"]
        p1, p2 = pairs[spec - 1]
        if nspec == 1:
            prior = [0.0, 1e5]
        else:
            prior = [-1e5, 1e5]
            for l in range(lo, hi + 1):
                D = Dl[l]
                if spec == 1:
                    prior[0] = max(prior[0], D[1] ** 2 / D[3])
                elif spec == 2:
                    prior[0] = max(prior[0], -np.sqrt(D[0] * D[3])); prior[1] = min(prior[1], np.sqrt(D[0] * D[3]))
                elif spec == 3:
                    prior[0] = max(prior[0], -np.sqrt(D[0] * D[5])); prior[1] = min(prior[1], np.sqrt(D[0] * D[5]))
                elif spec == 4:
                    prior[0] = max(prior[0], D[1] ** 2 / D[0])
                elif spec == 5:
                    prior[0] = max(prior[0], -np.sqrt(D[3] * D[5])); prior[1] = min(prior[1], np.sqrt(D[3] * D[5]))
                elif spec == 6:
                    prior[0] = max(prior[0], 0.0)
        d2 = Dl[lo, spec - 1]
        x_in = [max(d2 - 3 * bn["sigma"], 0.5 * (d2 + prior[0])), d2, min(d2 + 3 * bn["sigma"], 0.5 * (d2 + prior[1]))]

        def lnL(x):
            tot = 0.0
            for l in range(lo, hi + 1):
                S = S_mat[:, :, l].copy()
                S[p1, p2] = x / (l * (l + 1) / 2.0 / np.pi * RJ[p1] * RJ[p2])
                S[p2, p1] = S[p1, p2]
                for i in range(nmaps):
                    if S[i, i] == 0.0:
                        S[i, i] = 1.0
                Si, status, ln_det = _inv_chol(S)
                if status != 0:
                    return -1e30
                tot -= 0.5 * ((2 * l + 1) * ln_det + (2 * l + 1) * np.sum(sig[:, :, l] * Si.T))
            return tot

        s, status, _, took = sample_invsamp(uniforms[used], x_in, lnL, prior)
        used += 1 if took else 0
        if status != 0:
            return False, used
        Dl[lo:hi + 1, spec - 1] = s
    return True, used


def sample_cls_lookup(Dl, Dl_lookup, lmin_lookup, active, sigma_l_vec, S_mat, RJ2unit, eta):
    """sample_Dl_lookup (comm_Cl_mod.f90:1063-1145).  Dl_lookup[(nl), 6, nmodel]; active: 6 flags; Dl updated in place.
    Returns (ok, chosen 0-based model index)."""
    nl, nspec, n = Dl_lookup.shape
    pairs = spec_pairs(3)
    sig = sigma_l_matrix(np.asarray(sigma_l_vec), 3)
    RJ = np.asarray(RJ2unit, dtype=np.float64)
    lnL = np.zeros(n)
    for i in range(n):
        for l in range(lmin_lookup, lmin_lookup + nl):
            S = S_mat[:, :, l].copy()
            for m, (j, k) in enumerate(pairs):
                if active[m]:
                    S[j, k] = S[k, j] = Dl_lookup[l - lmin_lookup, m, i] / (l * (l + 1) / 2.0 / np.pi * RJ[j] * RJ[k])
            for j in range(3):
                if S[j, j] == 0.0:
                    S[j, j] = 1.0
            Si, status, ln_det = _inv_chol(S)
            if status != 0:
                lnL[i] = -1e30
            else:
                lnL[i] = lnL[i] - 0.5 * ((2 * l + 1) * ln_det + (2 * l + 1) * np.sum(sig[:, :, l] * Si.T))
    if np.all(lnL == -1e30):
        return False, -1
    P = np.where(lnL > -1e30, np.exp(lnL - lnL.max()), 0.0)
    P = P / P.sum()
    w, i = 0.0, 0
    while w < eta:
        w += P[i]
        i += 1
    i = max(i, 1)
    for l in range(lmin_lookup, lmin_lookup + nl):
        for m in range(6):
            if active[m]:
                Dl[l, m] = Dl_lookup[l - lmin_lookup, m, i - 1]
    return True, i - 1
