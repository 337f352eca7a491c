// Host side of the C_l Gibbs step that follows the amplitude solve (commander.f90:229 sample_powspec): the per-l
// signal-covariance tables (comm_Cl%updateS, comm_Cl_mod.f90:316-384) and the `binned` conditional sampler
// (sample_Cls_inverse_wishart2 + sample_InvSamp, comm_Cl_mod.f90:1008-1249, InvSamp_mod.f90:35-294).  sigma_l comes from
// the device (cmdr_sigma_l_dev); what is left is O(lmax) scalar work on matrices of order <= 3 that the reference runs
// on rank 0 only, so it stays on the host.  Uniform variates are supplied by the caller (the Fortran driver keeps its
// planck_rng handle), one per sampled bin, in the order the reference would draw them.
#include "cl_sampler.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"

namespace cmdr {
namespace {

constexpr int kMaxEvals = 1000;       // INVSAMP_MAX_NUM_EVALS
constexpr int kNSpline = 10000;       // N_SPLINE
constexpr double kDeltaLnL = 12.5;    // five sigma
constexpr double kTol = 1e-2;         // TOLERANCE

struct Mat3 {
    int n = 1;
    double a[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
};

// symmetric eigen-decomposition of order <= 3 by cyclic Jacobi rotations: A = V diag(w) V^T
void eig_sym(const Mat3& A, double w[3], double V[3][3]) {
    const int n = A.n;
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { a[i][j] = A.a[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += a[i][i] * a[i][i];
            for (int j = i + 1; j < n; ++j) off += a[i][j] * a[i][j];
        }
        if (off == 0.0 || off < 1e-34 * diag) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {   // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {   // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) w[i] = a[i][i];
}

// compute_hermitian_root without `trunc` (math_tools.f90:606-662): A(1,1) = -1e30 if an eigenvalue is <= 0
bool herm_root(Mat3& A, double pow) {
    double w[3], V[3][3];
    eig_sym(A, w, V);
    for (int i = 0; i < A.n; ++i)
        if (!(w[i] > 0.0)) { A.a[0][0] = -1e30; return false; }
    double f[3];
    for (int i = 0; i < A.n; ++i) f[i] = std::pow(w[i], pow);
    for (int i = 0; i < A.n; ++i)
        for (int j = 0; j < A.n; ++j) {
            double s = 0.0;
            for (int k = 0; k < A.n; ++k) s += V[i][k] * f[k] * V[j][k];
            A.a[i][j] = s;
        }
    return true;
}

// invert_matrix(S, cholesky=.true., status, ln_det) of math_tools.f90:76-152 for order <= 3
bool inv_chol(Mat3& S, double& ln_det) {
    const int n = S.n;
    double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < n; ++j) {
        double d = S.a[j][j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 0.0)) return false;
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double v = S.a[i][j];
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
    ln_det = 0.0;
    for (int i = 0; i < n; ++i) ln_det += 2.0 * std::log(L[i][i]);
    double Li[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // inverse of the lower factor
    for (int j = 0; j < n; ++j) {
        Li[j][j] = 1.0 / L[j][j];
        for (int i = j + 1; i < n; ++i) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v -= L[i][k] * Li[k][j];
            Li[i][j] = v / L[i][i];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double v = 0.0;
            for (int k = std::max(i, j); k < n; ++k) v += Li[k][i] * Li[k][j];
            S.a[i][j] = v;
        }
    return true;
}

int nmaps_to_nspec(int nmaps) { return nmaps * (nmaps + 1) / 2; }

// natural cubic spline second derivatives (spline_plain with yp1 = ypn = 1e30, spline_1D_mod.f90:109-149)
void spline_nat(const std::vector<double>& x, const std::vector<double>& y, std::vector<double>& y2) {
    const int n = (int)x.size();
    std::vector<double> a(n, 0.0), b(n, 0.0), c(n, 0.0), r(n, 0.0), gam(n, 0.0);
    for (int i = 0; i < n - 1; ++i) { c[i] = x[i + 1] - x[i]; r[i] = 6.0 * ((y[i + 1] - y[i]) / c[i]); }
    for (int i = n - 2; i >= 1; --i) r[i] -= r[i - 1];
    for (int i = 1; i < n - 1; ++i) { a[i] = c[i - 1]; b[i] = 2.0 * (c[i] + a[i]); }
    b[0] = b[n - 1] = 1.0;
    r[0] = c[0] = 0.0;
    r[n - 1] = a[n - 1] = 0.0;
    y2.assign(n, 0.0);
    double bet = b[0];
    y2[0] = r[0] / bet;
    for (int j = 1; j < n; ++j) {
        gam[j] = c[j - 1] / bet;
        bet = b[j] - a[j] * gam[j];
        y2[j] = (r[j] - a[j] * y2[j - 1]) / bet;
    }
    for (int j = n - 2; j >= 0; --j) y2[j] -= gam[j + 1] * y2[j + 1];
}

// splint_plain (spline_1D_mod.f90:151-172) with locate_dp (locate_mod.f90:69-102); klo is 0-based here
double splint(const std::vector<double>& xa, const std::vector<double>& ya, const std::vector<double>& y2, double x) {
    const int n = (int)xa.size();
    int loc;
    if (x == xa[0]) loc = 1;
    else if (x == xa[n - 1]) loc = n - 1;
    else loc = (int)(std::upper_bound(xa.begin(), xa.end(), x) - xa.begin());
    const int klo = std::max(std::min(loc, n - 1), 1) - 1, khi = klo + 1;
    const double h = xa[khi] - xa[klo];
    const double a = (xa[khi] - x) / h, b = (x - xa[klo]) / h;
    return a * ya[klo] + b * ya[khi] + ((a * a * a - a) * y2[klo] + (b * b * b - b) * y2[khi]) * (h * h) / 6.0;
}

struct Grid {
    std::vector<double> x, y;
    int stat = 0;
    void insert(double xn, double yn) {   // update_InvSamp_sample_set
        if ((int)x.size() == kMaxEvals) { ++stat; return; }
        size_t i = 0;
        while (i < x.size() && !(xn < x[i])) ++i;
        x.insert(x.begin() + i, xn);
        y.insert(y.begin() + i, yn);
    }
};

// sample_InvSamp (InvSamp_mod.f90:35-294), no precomputed grid, no optimize.  Returns status; `took` says whether the
// uniform variate was consumed (it is drawn only after the grid has converged).
template <typename LnL>
int inv_samp(double eta, const double x_in[3], LnL&& lnL, const double prior[2], double& sample, bool& took) {
    took = false;
    sample = 1e30;
    Grid g;
    for (int i = 0; i < 3; ++i) { g.x.push_back(x_in[i]); g.y.push_back(lnL(x_in[i])); }
    // bracket the peak
    while (g.y[0] > g.y[1] && g.x[0] > prior[0] && g.stat == 0) {
        const double xn = 0.5 * (g.x[0] + prior[0]);
        g.insert(xn, lnL(xn));
    }
    while (g.y.back() > g.y[g.y.size() - 2] && g.x.back() < prior[1] && g.stat == 0) {
        const size_t n = g.x.size();
        const double xn = std::min(g.x[n - 1] + 1.61803 * (g.x[n - 1] - g.x[n - 2]), prior[1]);
        g.insert(xn, lnL(xn));
    }
    if (g.stat != 0) return g.stat;
    // bound the five-sigma range
    double peak = *std::max_element(g.y.begin(), g.y.end());
    while (peak - g.y[0] < kDeltaLnL && g.x[0] > prior[0] && g.stat == 0) {
        const double xn = 0.5 * (g.x[0] + prior[0]);
        g.insert(xn, lnL(xn));
    }
    while (peak - g.y.back() < kDeltaLnL && g.x.back() < prior[1] && g.stat == 0) {
        const size_t n = g.x.size();
        const double xn = std::min(g.x[n - 1] + 1.61803 * (g.x[n - 1] - g.x[n - 2]), prior[1]);
        g.insert(xn, lnL(xn));
    }
    if (g.stat != 0) return g.stat;
    // refine until the spline predicts every new midpoint to `tol`
    std::vector<double> xs, ys, y2;
    double eps = 1e30;
    for (int iter = 1; eps > kTol; ++iter) {
        xs = g.x;
        ys = g.y;
        const int m = (int)xs.size();
        peak = *std::max_element(ys.begin(), ys.end());
        spline_nat(xs, ys, y2);
        eps = 0.0;
        for (int i = m - 1; i >= 1; --i) {   // intervals from the top: insertions never move the entries below
            if (peak - ys[i - 1] < kDeltaLnL || peak - ys[i] < kDeltaLnL) {
                const double xn = 0.5 * (xs[i - 1] + xs[i]);
                const double yn = lnL(xn), yp = splint(xs, ys, y2, xn);
                eps = std::max(std::fabs(yn - yp), eps);
                if (std::fabs(yn - yp) > kTol) g.insert(xn, yn);
            }
            if (g.stat != 0) break;
        }
        CMDR_REQUIRE(iter <= 100, "InvSamp: no convergence in 100 refinements (the reference stops here)");
        if (g.stat != 0) return g.stat;
    }
    const int n = (int)g.x.size();
    spline_nat(g.x, g.y, y2);
    peak = *std::max_element(g.y.begin(), g.y.end());
    int a = 0, b = n - 1;
    while (peak - g.y[a + 1] > kDeltaLnL && g.y[a + 1] > g.y[a]) ++a;
    while (peak - g.y[b - 1] > kDeltaLnL && g.y[b - 1] > g.y[b]) --b;
    const double x_min = g.x[a], x_max = g.x[b];
    const double dx = (x_max - x_min) / (kNSpline - 1.0);
    std::vector<double> x(kNSpline), P(kNSpline), F(kNSpline);
    double pmax = -1e300;
    for (int i = 0; i < kNSpline; ++i) {
        x[i] = x_min + dx * (double)i;
        P[i] = splint(g.x, g.y, y2, x[i]);
        pmax = std::max(pmax, P[i]);
    }
    for (int i = 0; i < kNSpline; ++i) P[i] = std::exp(P[i] - pmax);
    F[0] = 0.0;
    for (int j = 1; j < kNSpline; ++j) F[j] = F[j - 1] + dx * 0.5 * (P[j - 1] + P[j]);
    const double Fn = F[kNSpline - 1];
    for (int j = 0; j < kNSpline; ++j) F[j] /= Fn;
    took = true;
    int i = 1;
    while (eta > F[i] && i < kNSpline - 1) ++i;
    double s;
    if (i == kNSpline - 1) s = x[kNSpline - 1];
    else s = x[i - 1] + (eta - F[i - 1]) * (x[i] - x[i - 1]) / (F[i] - F[i - 1]);
    if (s != s) return 1;
    sample = std::max(std::min(s, prior[1]), prior[0]);
    return 0;
}

}  // namespace

// get_Cl_apod (comm_Cl_mod.f90:676-704)
double cl_apod(int l, int l_apod, int lmax, int lmax_prior, bool positive) {
    const double alpha = std::log(1e3);
    double f;
    if (l_apod > 0) {
        if (l <= l_apod) f = 1.0;
        else if (l > lmax) f = 0.0;
        else { const double r = (double)(l - l_apod) / (double)(lmax - l_apod + 1); f = std::exp(-alpha * r * r); }
    } else {
        const int la = std::abs(l_apod);
        if (l >= la) f = 1.0;
        else if (l == 0 || l > lmax) f = 0.0;
        else { const double r = (double)(la - l) / (double)(la - 1); f = std::exp(-alpha * r * r); }
    }
    if (lmax_prior >= 0 && l < lmax_prior) {
        const double c = 0.5 * (std::cos(M_PI * (double)(std::max(l, 1) - lmax_prior) / (double)lmax_prior) + 1.0);
        f *= c * c;
    }
    if (!positive && f != 0.0) f = 1.0 / f;
    return f;
}

// The apodisation enters matmulSqrtS / matmulS / matmulSqrtInvS / getCl as a per-l scalar (comm_Cl_mod.f90:550-674,
// 1440-1456): fold it into the tables the solver context receives.
void cl_apply_apod(int lmax, int nmaps, int l_apod, int lmax_prior, double* sqrtS, double* sqrtInvS, double* S) {
    CMDR_REQUIRE(lmax >= 0 && nmaps >= 1 && nmaps <= 3 && sqrtS && sqrtInvS && S, "bad arguments");
    for (int l = 0; l <= lmax; ++l) {
        const double f = cl_apod(l, l_apod, lmax, lmax_prior, true), g = cl_apod(l, l_apod, lmax, lmax_prior, false);
        for (int k = 0; k < nmaps * nmaps; ++k) {
            sqrtS[(size_t)nmaps * nmaps * l + k] *= f;
            S[(size_t)nmaps * nmaps * l + k] *= f * f;
            sqrtInvS[(size_t)nmaps * nmaps * l + k] *= g;
        }
    }
}

int cl_update_S(int lmax, int nmaps, int lmin, const double* Dl, const double* RJ2unit, double* sqrtS, double* sqrtInvS,
                double* S) {
    CMDR_REQUIRE(lmax >= 0 && nmaps >= 1 && nmaps <= 3 && Dl && RJ2unit && sqrtS && sqrtInvS && S, "bad arguments");
    const int64_t ld = lmax + 1;
    int nfail = 0;
    for (int l = 0; l <= lmax; ++l) {
        Mat3 M;
        M.n = nmaps;
        bool ok[3] = {true, true, true};
        int k = 0;
        for (int i = 0; i < nmaps; ++i)
            for (int j = i; j < nmaps; ++j, ++k) {
                const double D = Dl[l + ld * k];
                double v;
                if (l < lmin) v = 0.0;
                else if (l == 0) v = D;
                else v = D / ((double)l * (l + 1) / (2.0 * M_PI));
                v /= RJ2unit[i] * RJ2unit[j];
                M.a[i][j] = M.a[j][i] = v;
                if (i == j) ok[i] = D > 0.0;
            }
        auto blank = [&](Mat3& A, bool unit) {
            for (int i = 0; i < nmaps; ++i)
                if (!ok[i]) {
                    for (int j = 0; j < nmaps; ++j) A.a[i][j] = A.a[j][i] = 0.0;
                    if (unit) A.a[i][i] = 1.0;
                }
        };
        blank(M, true);
        Mat3 R = M, Ri = M;
        if (!herm_root(R, 0.5)) ++nfail;
        blank(R, false);
        herm_root(Ri, -0.5);
        blank(Ri, false);
        double* o1 = sqrtS + (size_t)nmaps * nmaps * l;
        double* o2 = sqrtInvS + (size_t)nmaps * nmaps * l;
        double* o3 = S + (size_t)nmaps * nmaps * l;
        for (int j = 0; j < nmaps; ++j)
            for (int i = 0; i < nmaps; ++i) {
                o1[i + nmaps * j] = R.a[i][j];
                o2[i + nmaps * j] = Ri.a[i][j];
                double s = 0.0;
                for (int q = 0; q < nmaps; ++q) s += R.a[i][q] * R.a[q][j];
                o3[i + nmaps * j] = s;
            }
    }
    return nfail;
}

int cl_sample_binned(int lmax, int nmaps, const double* sigma_l, const double* S_mat, const double* RJ2unit, int nbin,
                     const cmdr_cl_bin* bins, const double* uniform, int nuniform, double* Dl, int* nused) {
    CMDR_REQUIRE(lmax >= 0 && nmaps >= 1 && nmaps <= 3 && sigma_l && S_mat && RJ2unit && Dl && nbin >= 0 &&
                 (nbin == 0 || bins) && nuniform >= 0 && (nuniform == 0 || uniform), "bad arguments");
    const int64_t ld = lmax + 1;
    const int nspec = nmaps_to_nspec(nmaps);
    int pi_[6], pj_[6], k = 0;
    for (int i = 0; i < nmaps; ++i)
        for (int j = i; j < nmaps; ++j, ++k) { pi_[k] = i; pj_[k] = j; }
    // getSigmaL(sigma_l_mat) with the zero-diagonal fix of comm_Cl_mod.f90:1026-1030
    std::vector<Mat3> sig(lmax + 1);
    for (int l = 0; l <= lmax; ++l) {
        sig[l].n = nmaps;
        for (int q = 0; q < nspec; ++q) sig[l].a[pi_[q]][pj_[q]] = sig[l].a[pj_[q]][pi_[q]] = sigma_l[l + ld * q];
        for (int i = 0; i < nmaps; ++i)
            if (sig[l].a[i][i] == 0.0) sig[l].a[i][i] = 1.0;
    }
    int used = 0;
    auto D = [&](int l, int spec) -> double& { return Dl[l + ld * (spec - 1)]; };
    for (int ib = 0; ib < nbin; ++ib) {
        const cmdr_cl_bin& B = bins[ib];
        if (!B.sample) continue;
        CMDR_REQUIRE(B.lmin >= 0 && B.lmax <= lmax && B.lmin <= B.lmax && B.spec >= 1 && B.spec <= nspec, "bad C_l bin");
        double prior[2];
        if (nspec == 1) {
            prior[0] = 0.0;
            prior[1] = 1e5;
        } else {
            CMDR_REQUIRE(nspec == 6, "polarised C_l sampling needs nmaps = 3 (the reference indexes TT, TE, EE, BB)");
            prior[0] = -1e5;
            prior[1] = 1e5;
            for (int l = B.lmin; l <= B.lmax; ++l) {
                switch (B.spec) {
                    case 1: prior[0] = std::max(prior[0], D(l, 2) * D(l, 2) / D(l, 4)); break;
                    case 2: prior[0] = std::max(prior[0], -std::sqrt(D(l, 1) * D(l, 4)));
                            prior[1] = std::min(prior[1], std::sqrt(D(l, 1) * D(l, 4))); break;
                    case 3: prior[0] = std::max(prior[0], -std::sqrt(D(l, 1) * D(l, 6)));
                            prior[1] = std::min(prior[1], std::sqrt(D(l, 1) * D(l, 6))); break;
                    case 4: prior[0] = std::max(prior[0], D(l, 2) * D(l, 2) / D(l, 1)); break;
                    case 5: prior[0] = std::max(prior[0], -std::sqrt(D(l, 4) * D(l, 6)));
                            prior[1] = std::min(prior[1], std::sqrt(D(l, 4) * D(l, 6))); break;
                    default: prior[0] = std::max(prior[0], 0.0); break;
                }
            }
        }
        CMDR_REQUIRE(!(prior[1] < prior[0]), "InvSamp: upper prior is below the lower prior (the reference stops here)");
        const double d2 = D(B.lmin, B.spec);
        const double x_in[3] = {std::max(d2 - 3 * B.sigma, 0.5 * (d2 + prior[0])), d2,
                                std::min(d2 + 3 * B.sigma, 0.5 * (d2 + prior[1]))};
        const int p1 = pi_[B.spec - 1], p2 = pj_[B.spec - 1];
        auto lnL = [&](double x) {   // lnL_invWishart, comm_Cl_mod.f90:1210-1247
            double tot = 0.0;
            for (int l = B.lmin; l <= B.lmax; ++l) {
                Mat3 Sm;
                Sm.n = nmaps;
                const double* s = S_mat + (size_t)nmaps * nmaps * l;
                for (int j = 0; j < nmaps; ++j)
                    for (int i = 0; i < nmaps; ++i) Sm.a[i][j] = s[i + nmaps * j];
                Sm.a[p1][p2] = Sm.a[p2][p1] = x / ((double)l * (l + 1) / 2.0 / M_PI * RJ2unit[p1] * RJ2unit[p2]);
                for (int i = 0; i < nmaps; ++i)
                    if (Sm.a[i][i] == 0.0) Sm.a[i][i] = 1.0;
                double ln_det;
                if (!inv_chol(Sm, ln_det)) return -1e30;
                double tr = 0.0;
                for (int i = 0; i < nmaps; ++i)
                    for (int j = 0; j < nmaps; ++j) tr += sig[l].a[i][j] * Sm.a[j][i];
                tot -= 0.5 * ((double)(2 * l + 1) * ln_det + (double)(2 * l + 1) * tr);
            }
            return tot;
        };
        CMDR_REQUIRE(used < nuniform, "not enough uniform variates: one per sampled bin");
        double s;
        bool took;
        const int status = inv_samp(uniform[used], x_in, lnL, prior, s, took);
        if (took) ++used;
        if (status != 0) {
            if (nused) *nused = used;
            return 1;   // ok = .false.: the bins sampled so far keep their new values, as in the reference
        }
        for (int l = B.lmin; l <= B.lmax; ++l) D(l, B.spec) = s;
    }
    if (nused) *nused = used;
    return 0;
}

// sample_Dl_lookup (comm_Cl_mod.f90:1063-1145): pick one of nmodel tabulated spectra for lmin_lookup..lmax_lookup by its
// inverse-Wishart likelihood; polarised components only (the reference declares S(3,3)).
int cl_sample_lookup(int lmax, int lmin_lookup, int lmax_lookup, int nmodel, const double* Dl_lookup, const int* active,
                     const double* sigma_l, const double* S_mat, const double* RJ2unit, double uniform, double* Dl,
                     int* chosen) {
    CMDR_REQUIRE(lmax >= 0 && lmin_lookup >= 0 && lmax_lookup >= lmin_lookup && lmax_lookup <= lmax && nmodel >= 1 &&
                 Dl_lookup && active && sigma_l && S_mat && RJ2unit && Dl, "bad arguments");
    const int nmaps = 3, nspec = 6;
    const int64_t ld = lmax + 1, nl = lmax_lookup - lmin_lookup + 1;
    int pi_[6], pj_[6], k = 0;
    for (int i = 0; i < nmaps; ++i)
        for (int j = i; j < nmaps; ++j, ++k) { pi_[k] = i; pj_[k] = j; }
    std::vector<double> lnL(nmodel, 0.0);
    for (int im = 0; im < nmodel; ++im) {
        for (int l = lmin_lookup; l <= lmax_lookup; ++l) {
            Mat3 Sm;
            Sm.n = nmaps;
            const double* sm = S_mat + (size_t)9 * l;
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) Sm.a[i][j] = sm[i + 3 * j];
            for (int q = 0; q < nspec; ++q)
                if (active[q]) {
                    const double v = Dl_lookup[(l - lmin_lookup) + nl * (q + (int64_t)nspec * im)] /
                                     ((double)l * (l + 1) / 2.0 / M_PI * RJ2unit[pi_[q]] * RJ2unit[pj_[q]]);
                    Sm.a[pi_[q]][pj_[q]] = Sm.a[pj_[q]][pi_[q]] = v;
                }
            for (int i = 0; i < 3; ++i)
                if (Sm.a[i][i] == 0.0) Sm.a[i][i] = 1.0;
            double ln_det;
            if (!inv_chol(Sm, ln_det)) {
                lnL[im] = -1e30;               // assigned, not accumulated: later multipoles keep adding (:1094-1103)
            } else {
                double sig[3][3];
                for (int q = 0; q < nspec; ++q) sig[pi_[q]][pj_[q]] = sig[pj_[q]][pi_[q]] = sigma_l[l + ld * q];
                for (int i = 0; i < 3; ++i)
                    if (sig[i][i] == 0.0) sig[i][i] = 1.0;
                double tr = 0.0;
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) tr += sig[i][j] * Sm.a[j][i];
                lnL[im] -= 0.5 * ((double)(2 * l + 1) * ln_det + (double)(2 * l + 1) * tr);
            }
        }
    }
    bool all_bad = true;
    for (double v : lnL) all_bad = all_bad && v == -1e30;
    if (all_bad) return 1;
    const double mx = *std::max_element(lnL.begin(), lnL.end());
    double sum = 0.0;
    std::vector<double> P(nmodel);
    for (int i = 0; i < nmodel; ++i) { P[i] = lnL[i] > -1e30 ? std::exp(lnL[i] - mx) : 0.0; sum += P[i]; }
    for (double& v : P) v /= sum;
    double w = 0.0;
    int pick = 0;
    while (w < uniform && pick < nmodel) { w += P[pick]; ++pick; }   // 1-based model index, as the reference's i
    if (pick == 0) pick = 1;                                         // uniform == 0: the reference would read lnL(0)
    for (int l = lmin_lookup; l <= lmax_lookup; ++l)
        for (int q = 0; q < nspec; ++q)
            if (active[q]) Dl[l + ld * q] = Dl_lookup[(l - lmin_lookup) + nl * (q + (int64_t)nspec * (pick - 1))];
    if (chosen) *chosen = pick - 1;
    return 0;
}

}  // namespace cmdr
