/*
 * oracle/sht_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product).
 *
 * fp64 CPU restatement of the spherical-harmonic transforms the Commander3 CR solver reaches
 * through its libsharp2 FFI:
 *   - job types / calling convention: commander3/src/sharp.f90:8-14 (SHARP_YtW=0, Y=1, Yt=2, WY=3),
 *     :186-241 (sharp_execute_d: one column pointer per map, always SHARP_DP)
 *   - which job Commander issues per Stokes column: commander3/src/comm_map_mod.f90:437-579
 *     (exec_sharp_Y / _Yt / _YtW / _WY; T = spin 0, (Q,U) = one spin-2 call)
 *   - a_lm storage = m-major real-packed: commander3/src/comm_map_mod.f90:228-261 (index build),
 *     :1213-1246 (lm2i), :1497-1520 (get_alm: complex a_lm = (v(+m) + i v(-m))/sqrt(2))
 *   - ring weights W = 1 + weight_ring, analysis weight w_ring*4pi/Npix: comm_map_mod.f90:266-283
 *
 * PARITY UNPINNED against libsharp2: libsharp2 (cmake/project_instructions.cmake:93, "master" tarball, or the
 * copy in HEALPix 3.70) is not present in /root/reference nor in this image, and the reference holds no test
 * or golden vector for this path.  This file restates the *published* algorithm (per-ring FFT + per-m
 * Legendre recursion on HEALPix RING geometry) and is itself pinned by an independent brute-force direct sum
 * over scipy.special.sph_harm_y (oracle/bruteforce.py, tests/golden/).
 *
 * Deliberately simple: scaled three-term recursion with an explicit power-of-two exponent, phases stored
 * ring-major, ring transforms either by direct DFT (fft_mode=0, the independent check) or by radix-2 +
 * Bluestein FFT (fft_mode=1, used for the timed CPU baseline).
 */
#include <complex.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

#define JOB_YtW 0
#define JOB_Y 1
#define JOB_Yt 2
#define JOB_WY 3

static const double TWOPI = 6.283185307179586476925286766559;
static const double PI = 3.141592653589793238462643383279;

/* ---------------------------------------------------------------- HEALPix RING geometry */
/* ring = 1..4*nside-1, north to south (in_ring ordering of comm_map_mod.f90:198-226). */
typedef struct {
    int nphi;       /* pixels in ring */
    double z;       /* cos(theta) */
    double sth;     /* sin(theta) */
    double phi0;    /* azimuth of first pixel */
    int64_t start;  /* RING index of first pixel */
} ringinfo;

static ringinfo ring_info(int nside, int ring) {
    ringinfo r;
    int64_t N = nside, npix = 12 * N * N;
    int northring = ring > 2 * nside ? 4 * nside - ring : ring;
    double fN = (double)N;
    if (northring < nside) { /* polar cap */
        double i = (double)northring;
        double omz = i * i / (3.0 * fN * fN); /* 1 - z */
        r.z = 1.0 - omz;
        r.sth = sqrt(omz * (2.0 - omz));
        r.nphi = 4 * northring;
        r.phi0 = PI / (4.0 * i);
        r.start = 2 * (int64_t)northring * (northring - 1);
    } else { /* equatorial belt */
        r.z = 4.0 / 3.0 - 2.0 * (double)northring / (3.0 * fN);
        r.sth = sqrt((1.0 - r.z) * (1.0 + r.z));
        r.nphi = 4 * nside;
        r.phi0 = ((northring - nside) & 1) ? 0.0 : PI / (4.0 * fN);
        r.start = 2 * N * (N - 1) + 4 * N * (int64_t)(northring - nside);
    }
    if (ring != northring) { /* southern mirror */
        r.z = -r.z;
        r.start = npix - r.start - r.nphi;
    }
    return r;
}

void orc_ring_info(int nside, int ring, int* nphi, double* z, double* sth, double* phi0, int64_t* start) {
    ringinfo r = ring_info(nside, ring);
    *nphi = r.nphi; *z = r.z; *sth = r.sth; *phi0 = r.phi0; *start = r.start;
}

/* ---------------------------------------------------------------- a_lm layout (P=1: all m on this rank) */
/* comm_map_mod.f90:228-261: m=0 block of lmax+1 reals, then per m>0 (l=m..lmax) interleaved (+m,-m). */
static inline int64_t mind(int lmax, int m) {
    if (m == 0) return 0;
    /* (lmax+1) + sum_{k=1}^{m-1} 2(lmax-k+1) */
    return (int64_t)(lmax + 1) + 2 * ((int64_t)(m - 1) * (lmax + 1) - (int64_t)(m - 1) * m / 2);
}
int64_t orc_lm2i(int lmax, int l, int m) { /* comm_map_mod.f90:1213-1246 */
    int am = m < 0 ? -m : m;
    if (l > lmax || am > l) return -1;
    if (m == 0) return l;
    return mind(lmax, am) + 2 * (l - am) + (m < 0 ? 1 : 0);
}

/* ---------------------------------------------------------------- FFT (radix-2 + Bluestein), fft_mode=1 */
/* twiddles exp(2 pi i k / M), k < M/2, per power of two M, built once (the first version called cos / sin for every
   butterfly group of every transform: ~3 M libm calls per Bluestein ring, which dominated the whole oracle) */
static cplx* g_tw[40];
static const cplx* tw_get(int n) {
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    if (!g_tw[lg]) {
#pragma omp critical(orc_tw)
        if (!g_tw[lg]) {
            cplx* t = (cplx*)malloc(sizeof(cplx) * (n / 2 + 1));
            for (int k = 0; k < n / 2; ++k) {
                const double ang = TWOPI * (double)k / (double)n;
                t[k] = cos(ang) + I * sin(ang);
            }
            g_tw[lg] = t;
        }
    }
    return g_tw[lg];
}

static void fft_pow2(cplx* a, int n, int sign) { /* in place, unnormalised, exp(sign*2*pi*i*jk/n) */
    const cplx* tw = n > 1 ? tw_get(n) : NULL;
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { cplx t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        const int stride = n / len;
        for (int k = 0; k < half; ++k) {
            const cplx w = sign > 0 ? tw[k * stride] : conj(tw[k * stride]);   /* exp(sign 2 pi i k / len) */
            for (int i = k; i < n; i += len) {
                cplx u = a[i], v = a[i + half] * w;
                a[i] = u + v;
                a[i + half] = u - v;
            }
        }
    }
}

typedef struct {
    int n, M;      /* M = 0: n is a power of two, plain FFT */
    cplx* w;       /* chirp w_j = exp(+i*pi*j^2/n), j<n */
    cplx* chat;    /* FFT_M of conj chirp (sign +1 transform); the sign -1 one is its conjugate-symmetric twin */
} fftplan;

static int is_pow2(int n) { return (n & (n - 1)) == 0; }

static void plan_init(fftplan* p, int n) {
    p->n = n; p->M = 0; p->w = NULL; p->chat = NULL;
    if (is_pow2(n)) return;
    int M = 1;
    while (M < 2 * n - 1) M <<= 1;
    p->M = M;
    p->w = (cplx*)malloc(sizeof(cplx) * n);
    p->chat = (cplx*)calloc(M, sizeof(cplx));
    for (int j = 0; j < n; ++j) {
        int64_t q = ((int64_t)j * j) % (2 * (int64_t)n); /* exact phase reduction */
        double ang = PI * (double)q / (double)n;
        p->w[j] = cos(ang) + I * sin(ang);
    }
    p->chat[0] = conj(p->w[0]);
    for (int j = 1; j < n; ++j) { p->chat[j] = conj(p->w[j]); p->chat[M - j] = conj(p->w[j]); }
    fft_pow2(p->chat, M, -1);
}
static void plan_free(fftplan* p) { free(p->w); free(p->chat); }

/* the Bluestein plans of all ring lengths 4 i, i <= nside, kept for the most recent nside (built in parallel) */
static fftplan* g_plans;
static int g_plans_nside;
static const fftplan* plans_get(int nside) {
    if (g_plans_nside == nside) return g_plans;
    if (g_plans) { for (int i = 1; i <= g_plans_nside; ++i) plan_free(&g_plans[i]); free(g_plans); }
    g_plans = (fftplan*)malloc(sizeof(fftplan) * (nside + 1));
    for (int lg = 1; (1 << lg) <= 16 * nside; ++lg) (void)tw_get(1 << lg);   /* outside the parallel loop */
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 1; i <= nside; ++i) plan_init(&g_plans[i], 4 * i);
    g_plans_nside = nside;
    return g_plans;
}

/* y_k = sum_j x_j exp(sign*2*pi*i*jk/n); x,y length n; work length >= max(n,M) */
static void fft_any(const fftplan* p, const cplx* x, cplx* y, cplx* work, int sign) {
    int n = p->n;
    if (p->M == 0) {
        memcpy(y, x, sizeof(cplx) * n);
        fft_pow2(y, n, sign);
        return;
    }
    int M = p->M;
    /* sign=+1: exp(2 pi i jk/n) = w_j w_k conj(w_{k-j}).  sign=-1: conjugate everything. */
    for (int j = 0; j < n; ++j) work[j] = x[j] * (sign > 0 ? p->w[j] : conj(p->w[j]));
    for (int j = n; j < M; ++j) work[j] = 0;
    fft_pow2(work, M, -1);
    if (sign > 0) for (int j = 0; j < M; ++j) work[j] *= p->chat[j];
    else          for (int j = 0; j < M; ++j) work[j] *= conj(p->chat[(M - j) & (M - 1)]);
    fft_pow2(work, M, +1);
    double inv = 1.0 / (double)M;
    for (int k = 0; k < n; ++k) y[k] = work[k] * inv * (sign > 0 ? p->w[k] : conj(p->w[k]));
}

/* ---------------------------------------------------------------- Legendre helpers */
/* eps_lm = sqrt((l^2-m^2)/(4l^2-1)) */
static inline double eps_lm(int l, int m) {
    double dl = (double)l, dm = (double)m;
    return sqrt((dl * dl - dm * dm) / (4.0 * dl * dl - 1.0));
}

/* libsharp's (ring,m) cut: all lambda_lm(theta), l<=lmax, are negligible for m above this. */
static int mlim_of(int lmax, int spin, double sth, double cth) {
    double ofs = lmax * 0.01;
    if (ofs < 100.) ofs = 100.;
    double b = -2 * spin * fabs(cth);
    double t1 = lmax * sth + ofs;
    double c = (double)spin * spin - t1 * t1;
    double discr = b * b - 4 * c;
    if (discr <= 0) return lmax;
    double res = (-b + sqrt(discr)) / 2.;
    if (res > lmax) res = lmax;
    return (int)(res + 0.5);
}

#define RESCALE_BIG 0x1p+300
#define RESCALE_INV 0x1p-300

/* ring stage shared by the spin-0 and spin-2 transforms: direction 0 = pixels -> phases, 1 = phases -> pixels */
static void ring_stage(int synth, int nside, int lmax, const double* wring, int weighted, double* map, cplx* ph,
                       int fft_mode, const ringinfo* ri) {
    const int nring = 4 * nside - 1, mmax = lmax, nm = mmax + 1;
    const int64_t npix = 12 * (int64_t)nside * nside;
    const fftplan* plans = fft_mode ? plans_get(nside) : NULL;
    /* ---------------- analysis: pixels -> phases (per ring) */
    if (!synth) {
#pragma omp parallel
        {
            cplx* x = (cplx*)malloc(sizeof(cplx) * 4 * nside);
            cplx* y = (cplx*)malloc(sizeof(cplx) * 4 * nside);
            cplx* work = (cplx*)malloc(sizeof(cplx) * 16 * nside);
            double* ct = (double*)malloc(sizeof(double) * 4 * nside);
            double* st = (double*)malloc(sizeof(double) * 4 * nside);
#pragma omp for schedule(dynamic, 4)
            for (int r = 1; r <= nring; ++r) {
                const ringinfo R = ri[r];
                const int n = R.nphi;
                const int northring = r > 2 * nside ? 4 * nside - r : r;
                double wgt = 1.0;
                if (weighted) wgt = (wring ? wring[northring - 1] : 1.0) * 4.0 * PI / (double)npix;
                const double* pm = map + R.start;
                cplx* out = ph + (size_t)(r - 1) * nm;
                if (fft_mode) {
                    for (int k = 0; k < n; ++k) x[k] = pm[k] * wgt;
                    fft_any(&plans[n / 4], x, y, work, -1);
                    for (int m = 0; m <= mmax; ++m) {
                        double ang = -(double)m * R.phi0;
                        out[m] = y[m % n] * (cos(ang) + I * sin(ang));
                    }
                } else {
                    for (int j = 0; j < n; ++j) { ct[j] = cos(TWOPI * j / n); st[j] = sin(TWOPI * j / n); }
                    for (int m = 0; m <= mmax; ++m) {
                        double sr = 0, si = 0;
                        int64_t idx = 0, step = m % n;
                        for (int k = 0; k < n; ++k) {
                            sr += pm[k] * ct[idx];
                            si -= pm[k] * st[idx];
                            idx += step; if (idx >= n) idx -= n;
                        }
                        double ang = -(double)m * R.phi0;
                        out[m] = (sr + I * si) * wgt * (cos(ang) + I * sin(ang));
                    }
                }
            }
            free(x); free(y); free(work); free(ct); free(st);
        }
    }

    /* ---------------- synthesis: phases -> pixels (per ring) */
    if (synth) {
#pragma omp parallel
        {
            cplx* x = (cplx*)malloc(sizeof(cplx) * 4 * nside);
            cplx* y = (cplx*)malloc(sizeof(cplx) * 4 * nside);
            cplx* work = (cplx*)malloc(sizeof(cplx) * 16 * nside);
            double* ct = (double*)malloc(sizeof(double) * 4 * nside);
            double* st = (double*)malloc(sizeof(double) * 4 * nside);
#pragma omp for schedule(dynamic, 4)
            for (int r = 1; r <= nring; ++r) {
                const ringinfo R = ri[r];
                const int n = R.nphi;
                const int northring = r > 2 * nside ? 4 * nside - r : r;
                double wgt = 1.0;
                if (weighted) wgt = (wring ? wring[northring - 1] : 1.0) * 4.0 * PI / (double)npix;
                double* pm = map + R.start;
                const cplx* in = ph + (size_t)(r - 1) * nm;
                if (fft_mode) {
                    for (int k = 0; k < n; ++k) x[k] = 0;
                    for (int m = 0; m <= mmax; ++m) {
                        double ang = (double)m * R.phi0;
                        x[m % n] += in[m] * (cos(ang) + I * sin(ang));
                    }
                    fft_any(&plans[n / 4], x, y, work, +1);
                    for (int k = 0; k < n; ++k) pm[k] = creal(y[k]) * wgt;
                } else {
                    for (int j = 0; j < n; ++j) { ct[j] = cos(TWOPI * j / n); st[j] = sin(TWOPI * j / n); }
                    for (int k = 0; k < n; ++k) pm[k] = 0;
                    for (int m = 0; m <= mmax; ++m) {
                        double ang = (double)m * R.phi0;
                        cplx g = in[m] * (cos(ang) + I * sin(ang));
                        double gr = creal(g), gi = cimag(g);
                        int64_t idx = 0, step = m % n;
                        for (int k = 0; k < n; ++k) {
                            pm[k] += gr * ct[idx] - gi * st[idx];
                            idx += step; if (idx >= n) idx -= n;
                        }
                    }
                    if (wgt != 1.0) for (int k = 0; k < n; ++k) pm[k] *= wgt;
                }
            }
            free(x); free(y); free(work); free(ct); free(st);
        }
    }
}

/*
 * Scalar (spin-0) SHT on the full HEALPix sphere, one map.
 *   job      : JOB_Y / JOB_WY (alm -> map), JOB_Yt / JOB_YtW (map -> alm)
 *   wring    : [2*nside] ring weights W (comm_map_mod.f90:266-283 passes 1+weight_ring); NULL = 1
 *   alm      : real-packed m-major, (lmax+1)^2 doubles
 *   map      : RING-ordered, 12*nside^2 doubles
 *   fft_mode : 0 direct DFT per ring (independent check), 1 FFT
 *   use_mlim : 1 = skip (ring,m) pairs beyond libsharp's mlim
 * Returns 0.
 */
int orc_sht(int job, int nside, int lmax, const double* wring, double* alm, double* map, int fft_mode,
            int use_mlim, int nthreads) {
    const int nring = 4 * nside - 1, npair = 2 * nside, mmax = lmax, nm = mmax + 1;
    const int synth = (job == JOB_Y || job == JOB_WY);
    const int weighted = (job == JOB_YtW || job == JOB_WY);
    const double sqrt2 = sqrt(2.0);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
    cplx* ph = (cplx*)calloc((size_t)nring * nm, sizeof(cplx)); /* ph[(ring-1)*nm + m] */
    if (!ph) return -1;
    ringinfo* ri = (ringinfo*)malloc(sizeof(ringinfo) * (nring + 1));
    for (int r = 1; r <= nring; ++r) ri[r] = ring_info(nside, r);
    /* log lambda_mm prefactors: lambda_mm = (-1)^m sqrt((2m+1)!!/(4 pi (2m)!!)) sin^m */
    double* logpref = (double*)malloc(sizeof(double) * nm);
    logpref[0] = -0.5 * log(4.0 * PI);
    for (int m = 1; m <= mmax; ++m) logpref[m] = logpref[m - 1] + 0.5 * log((2.0 * m + 1.0) / (2.0 * m));

    if (!synth) ring_stage(0, nside, lmax, wring, weighted, map, ph, fft_mode, ri);

    /* ---------------- Legendre stage, parallel over m */
#pragma omp parallel
    {
        double* ieps = (double*)malloc(sizeof(double) * (lmax + 2));
        double* epsv = (double*)malloc(sizeof(double) * (lmax + 2));
        cplx* a = (cplx*)malloc(sizeof(cplx) * (lmax + 1));
#pragma omp for schedule(dynamic, 1)
        for (int mi = 0; mi <= mmax; ++mi) {
            /* interleave long (small m) and short (large m) columns */
            const int m = (mi & 1) ? mmax - mi / 2 : mi / 2;
            for (int l = m; l <= lmax; ++l) { epsv[l] = eps_lm(l, m); ieps[l] = l > m ? 1.0 / epsv[l] : 0.0; }
            const int64_t base = mind(lmax, m);
            const double mfac = m > 0 ? sqrt2 : 1.0;
            if (synth) {
                for (int l = m; l <= lmax; ++l)
                    a[l] = m == 0 ? alm[base + l] + 0 * I
                                  : (alm[base + 2 * (l - m)] + I * alm[base + 2 * (l - m) + 1]) * mfac;
            } else {
                for (int l = m; l <= lmax; ++l) a[l] = 0;
            }
            for (int rp = 1; rp <= npair; ++rp) {
                const ringinfo R = ri[rp];
                const double x = R.z, sth = R.sth;
                if (use_mlim && m > mlim_of(lmax, 0, sth, x)) continue;
                const int has_south = rp < npair;
                /* lambda_mm as mantissa * 2^e */
                double l2 = (logpref[m] + (m > 0 ? (double)m * log(sth) : 0.0)) / M_LN2;
                double fl = floor(l2);
                long e = (long)fl;
                double lc = exp2(l2 - fl);
                if (m & 1) lc = -lc;
                double lp = 0.0;
                double sf = (e < -900) ? 0.0 : ldexp(1.0, (int)e);
                cplx Gn = 0, Gs = 0, Ge = 0, Go = 0;
                if (!synth) {
                    Gn = ph[(size_t)(rp - 1) * nm + m];
                    Gs = has_south ? ph[(size_t)(4 * nside - rp - 1) * nm + m] : 0;
                    Ge = Gn + Gs; Go = Gn - Gs; /* parity-even / parity-odd ring combinations */
                }
                cplx Fe = 0, Fo = 0;
                for (int l = m;; ++l) {
                    if (sf != 0.0) {
                        double lam = lc * sf;
                        if (synth) { if ((l - m) & 1) Fo += a[l] * lam; else Fe += a[l] * lam; }
                        else       { a[l] += (((l - m) & 1) ? Go : Ge) * lam; }
                    }
                    if (l == lmax) break;
                    double ln = (x * lc - epsv[l] * lp) * ieps[l + 1];
                    lp = lc; lc = ln;
                    if (fabs(lc) > RESCALE_BIG) {
                        lc *= RESCALE_INV; lp *= RESCALE_INV; e += 300;
                        sf = (e < -900) ? 0.0 : ldexp(1.0, (int)e);
                    }
                }
                if (synth) {
                    ph[(size_t)(rp - 1) * nm + m] = Fe + Fo;
                    if (has_south) ph[(size_t)(4 * nside - rp - 1) * nm + m] = Fe - Fo;
                }
            }
            if (!synth) {
                if (m == 0) for (int l = 0; l <= lmax; ++l) alm[base + l] = creal(a[l]);
                else for (int l = m; l <= lmax; ++l) {
                    alm[base + 2 * (l - m)] = creal(a[l]) * mfac;
                    alm[base + 2 * (l - m) + 1] = cimag(a[l]) * mfac;
                }
            }
        }
        free(ieps); free(epsv); free(a);
    }

    if (synth) ring_stage(1, nside, lmax, wring, weighted, map, ph, fft_mode, ri);
    free(logpref); free(ri); free(ph);
    return 0;
}

/*
 * Harmonic-space diagonal of Y^T N^-1 Y as Commander's compute_invN_lm defines it
 * (commander3/src/comm_N_mod.f90:127-197):
 *   N_lm = (-1)^m (2l+1)/sqrt(4pi) * Npix/4pi * sum_{l'<=lmax} a_l'0 sqrt(2l'+1) (l l l'; -m m 0)(l l l'; 0 0 0)
 * with a_l'0 the m=0 coefficients of YtW(siN^2) (":134 call invN_diag%YtW_scalar").  By the Gaunt integral
 * this equals  Npix/4pi * Int |Y_lm|^2 g dOmega  with  g(theta) = sum_{l'<=lmax} a_l'0 Y_l'0(theta),  which is
 * evaluated here exactly by Gauss-Legendre quadrature (integrand is a polynomial of degree <= 3*lmax in cos theta).
 * oracle/wigner.py evaluates the 3j form literally for small lmax and tests/ pin the two against each other.
 *   al0 : [lmax+1] the a_l0 of YtW(siN^2);  out : real-packed (lmax+1)^2, both (+m,-m) slots get the same value
 */
static void gauss_legendre(int n, double* x, double* w) {
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = cos(PI * (i + 0.75) / (n + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j; }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            double z1 = z; z = z1 - p1 / pp;
            if (fabs(z - z1) < 1e-16) break;
        }
        x[i] = z; x[n - 1 - i] = -z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

int orc_invn_diag(int nside, int lmax, const double* al0, double* out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
    const int ng = (3 * lmax) / 2 + 2;
    const double npix = 12.0 * nside * nside;
    double* gx = (double*)malloc(sizeof(double) * ng);
    double* gw = (double*)malloc(sizeof(double) * ng);
    double* g = (double*)malloc(sizeof(double) * ng);
    gauss_legendre(ng, gx, gw);
    /* g(theta_k) = sum_l' a_l'0 Y_l'0 */
    for (int k = 0; k < ng; ++k) {
        double x = gx[k], lp = 0.0, lc = sqrt(1.0 / (4.0 * PI)), s = 0;
        for (int l = 0;; ++l) {
            s += al0[l] * lc;
            if (l == lmax) break;
            double ln = (x * lc - eps_lm(l, 0) * lp) / eps_lm(l + 1, 0);
            lp = lc; lc = ln;
        }
        g[k] = s;
    }
    double* logpref = (double*)malloc(sizeof(double) * (lmax + 1));
    logpref[0] = -0.5 * log(4.0 * PI);
    for (int m = 1; m <= lmax; ++m) logpref[m] = logpref[m - 1] + 0.5 * log((2.0 * m + 1.0) / (2.0 * m));
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (lmax + 1));
        double* epsv = (double*)malloc(sizeof(double) * (lmax + 2));
#pragma omp for schedule(dynamic, 1)
        for (int m = 0; m <= lmax; ++m) {
            for (int l = m; l <= lmax + 1; ++l) epsv[l] = eps_lm(l, m);
            for (int l = m; l <= lmax; ++l) acc[l] = 0;
            for (int k = 0; k < ng; ++k) {
                double x = gx[k], sth = sqrt((1.0 - x) * (1.0 + x));
                double l2 = (logpref[m] + (m > 0 ? (double)m * log(sth) : 0.0)) / M_LN2;
                double fl = floor(l2);
                long e = (long)fl;
                double lc = exp2(l2 - fl), lp = 0.0;
                double sf = (e < -900) ? 0.0 : ldexp(1.0, (int)e);
                double wk = gw[k] * g[k] * TWOPI * npix / (4.0 * PI);
                for (int l = m;; ++l) {
                    if (sf != 0.0) { double lam = lc * sf; acc[l] += wk * lam * lam; }
                    if (l == lmax) break;
                    double ln = (x * lc - epsv[l] * lp) / epsv[l + 1];
                    lp = lc; lc = ln;
                    if (fabs(lc) > RESCALE_BIG) {
                        lc *= RESCALE_INV; lp *= RESCALE_INV; e += 300;
                        sf = (e < -900) ? 0.0 : ldexp(1.0, (int)e);
                    }
                }
            }
            int64_t base = mind(lmax, m);
            if (m == 0) for (int l = 0; l <= lmax; ++l) out[base + l] = acc[l];
            else for (int l = m; l <= lmax; ++l) { out[base + 2 * (l - m)] = acc[l]; out[base + 2 * (l - m) + 1] = acc[l]; }
        }
        free(acc); free(epsv);
    }
    free(logpref); free(gx); free(gw); free(g);
    return 0;
}

/* ======================================================================================= spin-2 (Q,U <-> E,B)
 * What Commander issues for the polarisation columns: one spin-2 call on alm(:,2:3) / map(:,2:3)
 * (commander3/src/comm_map_mod.f90:446-449, :519-523, :549-553) -- "COSMO" convention (POLCCONV, :1002):
 *     (Q +- iU)(p) = sum_lm a_{+-2,lm} +-2Y_lm(p),    a_{+-2,lm} = -(E_lm +- i B_lm)            
 * With W = (2lam + -2lam)/2, X = (2lam - -2lam)/2 this is the HEALPix form
 *     F^Q_m = -sum_l (E W + i B X),   F^U_m = -sum_l (B W - i E X),   map = Re F_0 + 2 Re sum_{m>0} F_m e^{i m phi}.
 * The spin-weighted lambdas follow the three-term recursion (pinned numerically against the Goldberg closed form,
 * oracle/bruteforce.py::spin_Y):
 *     s_lam_{l+1} = [ (x + s m / (l(l+1))) s_lam_l - C_l s_lam_{l-1} ] / C_{l+1},
 *     C_l = sqrt((l^2-m^2)(l^2-s^2) / (l^2 (4l^2-1))),   start l0 = max(m, 2),   s_lam(pi-theta) = (-1)^{l+m} (-s)_lam(theta).
 */
static double slam_start(int s, int m, double cth2, double sth2, double* log2mag) {
    /* s_lam_{l0,m}(theta), l0 = max(m,2), via the Goldberg sum (l0 <= m+2 terms only matter for m < 2);
       returns the sign, *log2mag = log2|value| (cth2 = cos(theta/2), sth2 = sin(theta/2)) */
    const int l = m > 2 ? m : 2;
    if (m >= 2) {
        /* (-1)^m sqrt((2m+1)/4pi (2m)!/((m+s)!(m-s)!)) cos^{m-s} sin^{m+s} */
        double lg = 0.5 * (log(2.0 * m + 1.0) - log(4.0 * PI) + lgamma(2.0 * m + 1.0) - lgamma(m + s + 1.0) - lgamma(m - s + 1.0));
        lg += (m - s) * log(cth2) + (m + s) * log(sth2);
        *log2mag = lg / M_LN2;
        return (m & 1) ? -1.0 : 1.0;
    }
    /* m = 0, 1: explicit Goldberg sum at l = 2 */
    double fact[8] = {1, 1, 2, 6, 24, 120, 720, 5040};
    double pref = ((m & 1) ? -1.0 : 1.0) * sqrt((2 * l + 1) / (4 * PI) * fact[l + m] * fact[l - m] / (fact[l + s] * fact[l - s]));
    double acc = 0;
    for (int r = 0; r <= l - s; ++r) {
        int k = r + s - m;
        if (k < 0 || k > l + s) continue;
        double c1 = fact[l - s] / (fact[r] * fact[l - s - r]), c2 = fact[l + s] / (fact[k] * fact[l + s - k]);
        acc += c1 * c2 * (((l - r - s) & 1) ? -1.0 : 1.0) * pow(sth2, 2 * l - 2 * r - s + m) * pow(cth2, 2 * r + s - m);
    }
    double v = pref * acc;
    if (v == 0.0) { *log2mag = -1e30; return 1.0; }
    *log2mag = log2(fabs(v));
    return v < 0 ? -1.0 : 1.0;
}

typedef struct { double lc, lp, sf; long e; } chain;

static void chain_init(chain* c, double sign, double l2) {
    if (l2 < -1e20) { c->lc = 0; c->lp = 0; c->e = 0; c->sf = 1.0; return; }
    double fl = floor(l2);
    c->e = (long)fl;
    c->lc = sign * exp2(l2 - fl);
    c->lp = 0.0;
    c->sf = (c->e < -900) ? 0.0 : ldexp(1.0, (int)c->e);
}
static inline void chain_step(chain* c, double coef, double cl, double icl1) {
    double ln = (coef * c->lc - cl * c->lp) * icl1;
    c->lp = c->lc; c->lc = ln;
    if (fabs(c->lc) > RESCALE_BIG) {
        c->lc *= RESCALE_INV; c->lp *= RESCALE_INV; c->e += 300;
        c->sf = (c->e < -900) ? 0.0 : ldexp(1.0, (int)c->e);
    }
}

int orc_sht_spin2(int job, int nside, int lmax, const double* wring, double* almE, double* almB, double* mapQ,
                  double* mapU, int fft_mode, int use_mlim, int nthreads) {
    const int nring = 4 * nside - 1, npair = 2 * nside, mmax = lmax, nm = mmax + 1;
    const int synth = (job == JOB_Y || job == JOB_WY);
    const int weighted = (job == JOB_YtW || job == JOB_WY);
    const double sqrt2 = sqrt(2.0);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
    cplx* phQ = (cplx*)calloc((size_t)nring * nm, sizeof(cplx));
    cplx* phU = (cplx*)calloc((size_t)nring * nm, sizeof(cplx));
    if (!phQ || !phU) return -1;
    ringinfo* ri = (ringinfo*)malloc(sizeof(ringinfo) * (nring + 1));
    for (int r = 1; r <= nring; ++r) ri[r] = ring_info(nside, r);
    if (!synth) {
        ring_stage(0, nside, lmax, wring, weighted, mapQ, phQ, fft_mode, ri);
        ring_stage(0, nside, lmax, wring, weighted, mapU, phU, fft_mode, ri);
    }
#pragma omp parallel
    {
        double* Cl = (double*)malloc(sizeof(double) * (lmax + 3));
        cplx* aE = (cplx*)malloc(sizeof(cplx) * (lmax + 1));
        cplx* aB = (cplx*)malloc(sizeof(cplx) * (lmax + 1));
#pragma omp for schedule(dynamic, 1)
        for (int mi = 0; mi <= mmax; ++mi) {
            const int m = (mi & 1) ? mmax - mi / 2 : mi / 2;
            const int l0 = m > 2 ? m : 2;
            if (l0 > lmax) continue;
            for (int l = l0; l <= lmax + 1; ++l) {
                double dl = l, dm = m;
                Cl[l] = sqrt((dl * dl - dm * dm) * (dl * dl - 4.0) / (dl * dl * (4.0 * dl * dl - 1.0)));
            }
            const int64_t base = mind(lmax, m);
            const double mfac = m > 0 ? sqrt2 : 1.0;
            for (int l = 0; l <= lmax; ++l) { aE[l] = 0; aB[l] = 0; }
            if (synth)
                for (int l = l0; l <= lmax; ++l) {
                    if (m == 0) { aE[l] = almE[base + l]; aB[l] = almB[base + l]; }
                    else {
                        aE[l] = (almE[base + 2 * (l - m)] + I * almE[base + 2 * (l - m) + 1]) * mfac;
                        aB[l] = (almB[base + 2 * (l - m)] + I * almB[base + 2 * (l - m) + 1]) * mfac;
                    }
                }
            for (int rp = 1; rp <= npair; ++rp) {
                const ringinfo R = ri[rp];
                const double x = R.z, sth = R.sth;
                if (use_mlim && m > mlim_of(lmax, 2, sth, x)) continue;
                const int has_south = rp < npair;
                const double th = atan2(sth, x), c2 = cos(0.5 * th), s2 = sin(0.5 * th);
                chain cp, cm;  /* +2 and -2 chains */
                double lg, sg;
                sg = slam_start(2, m, c2, s2, &lg);  chain_init(&cp, sg, lg);
                sg = slam_start(-2, m, c2, s2, &lg); chain_init(&cm, sg, lg);
                cplx GQn = 0, GQs = 0, GUn = 0, GUs = 0;
                if (!synth) {
                    GQn = phQ[(size_t)(rp - 1) * nm + m]; GUn = phU[(size_t)(rp - 1) * nm + m];
                    if (has_south) { GQs = phQ[(size_t)(4 * nside - rp - 1) * nm + m]; GUs = phU[(size_t)(4 * nside - rp - 1) * nm + m]; }
                }
                /* accumulators: "same-sign" (keeps sign in the south) and "flip" parts */
                cplx Qk = 0, Qf = 0, Uk = 0, Uf = 0;
                for (int l = l0;; ++l) {
                    const double lp2 = cp.lc * cp.sf, lm2 = cm.lc * cm.sf;
                    if (lp2 != 0.0 || lm2 != 0.0) {
                        const double W = 0.5 * (lp2 + lm2), X = 0.5 * (lp2 - lm2);
                        const int odd = (l + m) & 1;  /* W has parity (-1)^{l+m}, X the opposite */
                        if (synth) {
                            cplx qW = -aE[l] * W, qX = -I * aB[l] * X, uW = -aB[l] * W, uX = I * aE[l] * X;
                            if (!odd) { Qk += qW; Qf += qX; Uk += uW; Uf += uX; }
                            else      { Qf += qW; Qk += qX; Uf += uW; Uk += uX; }
                        } else {
                            /* transpose: north + south with the parity sign */
                            const double sW = odd ? -1.0 : 1.0, sX = -sW;
                            cplx gQW = GQn + sW * GQs, gQX = GQn + sX * GQs, gUW = GUn + sW * GUs, gUX = GUn + sX * GUs;
                            /* E: -W GQ  + conj-transpose of (i X) on U-phase: F^U += i E X  =>  E += -i X GU ... as real maps:
                               F^Q = -E W - i B X ; F^U = -B W + i E X.  Adjoint w.r.t. the real inner product of (re,im):
                               E += -W GQ + conj(i) X GU = -W GQ - i X GU ;  B += -W GU + i X GQ */
                            aE[l] += -W * gQW - I * X * gUX;
                            aB[l] += -W * gUW + I * X * gQX;
                        }
                    }
                    if (l == lmax) break;
                    const double sm = 2.0 * m / ((double)l * (l + 1.0));
                    chain_step(&cp, x + sm, l > l0 ? Cl[l] : 0.0, 1.0 / Cl[l + 1]);
                    chain_step(&cm, x - sm, l > l0 ? Cl[l] : 0.0, 1.0 / Cl[l + 1]);
                }
                if (synth) {
                    phQ[(size_t)(rp - 1) * nm + m] = Qk + Qf;
                    phU[(size_t)(rp - 1) * nm + m] = Uk + Uf;
                    if (has_south) {
                        phQ[(size_t)(4 * nside - rp - 1) * nm + m] = Qk - Qf;
                        phU[(size_t)(4 * nside - rp - 1) * nm + m] = Uk - Uf;
                    }
                }
            }
            if (!synth) {
                for (int l = 0; l <= lmax; ++l) {
                    if (l < m) continue;
                    if (m == 0) { almE[base + l] = creal(aE[l]); almB[base + l] = creal(aB[l]); }
                    else {
                        almE[base + 2 * (l - m)] = creal(aE[l]) * mfac; almE[base + 2 * (l - m) + 1] = cimag(aE[l]) * mfac;
                        almB[base + 2 * (l - m)] = creal(aB[l]) * mfac; almB[base + 2 * (l - m) + 1] = cimag(aB[l]) * mfac;
                    }
                }
            }
        }
        free(Cl); free(aE); free(aB);
    }
    if (!synth) { /* columns m > lmax never run; l < 2 entries stay as initialised by the caller: zero them */
        for (int64_t i = 0; i < (int64_t)(lmax + 1) * (lmax + 1); ++i) { (void)i; }
    }
    if (synth) {
        ring_stage(1, nside, lmax, wring, weighted, mapQ, phQ, fft_mode, ri);
        ring_stage(1, nside, lmax, wring, weighted, mapU, phU, fft_mode, ri);
    }
    free(ri); free(phQ); free(phU);
    return 0;
}

/* ======================================================================================= vectorised Legendre stages
 * The same transforms with the Legendre stage blocked over ring pairs: NV ring pairs advance together in explicit SIMD
 * vectors (GCC vector extensions, 4 doubles each; l stays sequential).  Identical arithmetic per ring pair -- recursion,
 * 2^e scaling, mlim cut, parity split -- except that (i) the order in which ring pairs are summed into a_lm differs and
 * (ii) the 2^300 rescale is looked for every 8th l instead of every l (the mantissa grows by < 4 per step, so it stays
 * far inside the double range; scaling by powers of two is exact).  Used as the default oracle and as the in-run CPU
 * baseline of bench.py ("same algorithm class as libsharp2"); tests/test_oracle.py pins it
 * against the plain loops above and against the brute-force goldens.  TEST INFRASTRUCTURE like the rest of this file. */
typedef double v4d __attribute__((vector_size(32), aligned(32)));
typedef long long v4l __attribute__((vector_size(32), aligned(32)));
#define VL 4                 /* doubles per vector */
#define NVV 4                /* vectors per block, scalar transform: 16 ring pairs */
#define NVV2 2               /* spin-2: 8 ring pairs */
typedef union { v4d v; double s[VL]; } u4d;
typedef union { v4l v; long long s[VL]; } u4l;
static inline v4d vbc(double a) { return (v4d){a, a, a, a}; }
static inline double vsum(v4d a) { u4d u; u.v = a; return (u.s[0] + u.s[1]) + (u.s[2] + u.s[3]); }

int orc_sht_fast(int job, int nside, int lmax, const double* wring, double* alm, double* map, int fft_mode,
                 int use_mlim, int nthreads) {
    const int nring = 4 * nside - 1, npair = 2 * nside, mmax = lmax, nm = mmax + 1;
    const int synth = (job == JOB_Y || job == JOB_WY);
    const int weighted = (job == JOB_YtW || job == JOB_WY);
    const double sqrt2 = sqrt(2.0);
    enum { NVs = VL * NVV };
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
    cplx* ph = (cplx*)calloc((size_t)nring * nm, sizeof(cplx));
    if (!ph) return -1;
    ringinfo* ri = (ringinfo*)malloc(sizeof(ringinfo) * (nring + 1));
    for (int r = 1; r <= nring; ++r) ri[r] = ring_info(nside, r);
    double* logpref = (double*)malloc(sizeof(double) * nm);
    logpref[0] = -0.5 * log(4.0 * PI);
    for (int m = 1; m <= mmax; ++m) logpref[m] = logpref[m - 1] + 0.5 * log((2.0 * m + 1.0) / (2.0 * m));
    if (!synth) ring_stage(0, nside, lmax, wring, weighted, map, ph, fft_mode, ri);
#pragma omp parallel
    {
        double* ieps = (double*)malloc(sizeof(double) * (lmax + 2));
        double* epsv = (double*)malloc(sizeof(double) * (lmax + 2));
        double* are = (double*)malloc(sizeof(double) * (lmax + 1));
        double* aim = (double*)malloc(sizeof(double) * (lmax + 1));
#pragma omp for schedule(dynamic, 1)
        for (int mi = 0; mi <= mmax; ++mi) {
            const int m = (mi & 1) ? mmax - mi / 2 : mi / 2;
            for (int l = m; l <= lmax; ++l) { epsv[l] = eps_lm(l, m); ieps[l] = l > m ? 1.0 / epsv[l] : 0.0; }
            ieps[lmax + 1] = 0.0;
            const int64_t base = mind(lmax, m);
            const double mfac = m > 0 ? sqrt2 : 1.0;
            for (int l = m; l <= lmax; ++l) {
                if (synth) {
                    are[l] = (m == 0 ? alm[base + l] : alm[base + 2 * (l - m)] * mfac);
                    aim[l] = (m == 0 ? 0.0 : alm[base + 2 * (l - m) + 1] * mfac);
                } else are[l] = aim[l] = 0.0;
            }
            for (int rp0 = 1; rp0 <= npair; rp0 += NVs) {
                u4d x[NVV], lc[NVV], lp[NVV], sf[NVV], Ger[NVV], Gei[NVV], Gor[NVV], Goi[NVV];
                v4d Fer[NVV], Fei[NVV], For[NVV], Foi[NVV];
                long e[NVs];
                int on[NVs], any = 0;
                for (int v = 0; v < NVs; ++v) {
                    const int rp = rp0 + v, k = v / VL, j = v % VL;
                    x[k].s[j] = lc[k].s[j] = lp[k].s[j] = sf[k].s[j] = 0;
                    Ger[k].s[j] = Gei[k].s[j] = Gor[k].s[j] = Goi[k].s[j] = 0;
                    e[v] = 0; on[v] = 0;
                    if (rp > npair) continue;
                    const ringinfo R = ri[rp];
                    if (use_mlim && m > mlim_of(lmax, 0, R.sth, R.z)) continue;
                    on[v] = 1; any = 1;
                    x[k].s[j] = R.z;
                    const double l2 = (logpref[m] + (m > 0 ? (double)m * log(R.sth) : 0.0)) / M_LN2;
                    const double fl = floor(l2);
                    e[v] = (long)fl;
                    lc[k].s[j] = (m & 1) ? -exp2(l2 - fl) : exp2(l2 - fl);
                    sf[k].s[j] = (e[v] < -900) ? 0.0 : ldexp(1.0, (int)e[v]);
                    if (!synth) {
                        const cplx Gn = ph[(size_t)(rp - 1) * nm + m];
                        const cplx Gs = rp < npair ? ph[(size_t)(4 * nside - rp - 1) * nm + m] : 0;
                        Ger[k].s[j] = creal(Gn + Gs); Gei[k].s[j] = cimag(Gn + Gs);
                        Gor[k].s[j] = creal(Gn - Gs); Goi[k].s[j] = cimag(Gn - Gs);
                    }
                }
                if (!any) continue;
                for (int k = 0; k < NVV; ++k) Fer[k] = Fei[k] = For[k] = Foi[k] = vbc(0.0);
                const v4d big = vbc(RESCALE_BIG), nbig = vbc(-RESCALE_BIG);
                for (int lb = m; lb <= lmax; lb += 8) {
                    const int le = lb + 8 <= lmax + 1 ? lb + 8 : lmax + 1;
                    v4l over = {0, 0, 0, 0};
                    for (int l = lb; l < le; ++l) {
                        const int odd = (l - m) & 1;
                        const v4d el = vbc(epsv[l]), ie1 = vbc(ieps[l + 1]);
                        if (synth) {
                            const v4d ar = vbc(are[l]), ai = vbc(aim[l]);
                            for (int k = 0; k < NVV; ++k) {
                                const v4d lam = lc[k].v * sf[k].v;
                                if (odd) { For[k] += ar * lam; Foi[k] += ai * lam; }
                                else     { Fer[k] += ar * lam; Fei[k] += ai * lam; }
                                const v4d ln = (x[k].v * lc[k].v - el * lp[k].v) * ie1;
                                lp[k].v = lc[k].v;
                                lc[k].v = ln;
                                over |= (ln > big) | (ln < nbig);
                            }
                        } else {
                            v4d sr = vbc(0.0), si = vbc(0.0);
                            for (int k = 0; k < NVV; ++k) {
                                const v4d lam = lc[k].v * sf[k].v;
                                sr += (odd ? Gor[k].v : Ger[k].v) * lam;
                                si += (odd ? Goi[k].v : Gei[k].v) * lam;
                                const v4d ln = (x[k].v * lc[k].v - el * lp[k].v) * ie1;
                                lp[k].v = lc[k].v;
                                lc[k].v = ln;
                                over |= (ln > big) | (ln < nbig);
                            }
                            are[l] += vsum(sr);
                            aim[l] += vsum(si);
                        }
                    }
                    u4l ov; ov.v = over;
                    if (ov.s[0] | ov.s[1] | ov.s[2] | ov.s[3])
                        for (int v = 0; v < NVs; ++v) {
                            const int k = v / VL, j = v % VL;
                            if (fabs(lc[k].s[j]) > RESCALE_BIG) {
                                lc[k].s[j] *= RESCALE_INV; lp[k].s[j] *= RESCALE_INV; e[v] += 300;
                                sf[k].s[j] = (e[v] < -900) ? 0.0 : ldexp(1.0, (int)e[v]);
                            }
                        }
                }
                if (synth)
                    for (int v = 0; v < NVs; ++v) {
                        const int rp = rp0 + v, k = v / VL, j = v % VL;
                        if (!on[v]) continue;
                        u4d er, ei, or_, oi; er.v = Fer[k]; ei.v = Fei[k]; or_.v = For[k]; oi.v = Foi[k];
                        ph[(size_t)(rp - 1) * nm + m] = (er.s[j] + or_.s[j]) + I * (ei.s[j] + oi.s[j]);
                        if (rp < npair) ph[(size_t)(4 * nside - rp - 1) * nm + m] = (er.s[j] - or_.s[j]) + I * (ei.s[j] - oi.s[j]);
                    }
            }
            if (!synth) {
                if (m == 0) for (int l = 0; l <= lmax; ++l) alm[base + l] = are[l];
                else for (int l = m; l <= lmax; ++l) {
                    alm[base + 2 * (l - m)] = are[l] * mfac;
                    alm[base + 2 * (l - m) + 1] = aim[l] * mfac;
                }
            }
        }
        free(ieps); free(epsv); free(are); free(aim);
    }
    if (synth) ring_stage(1, nside, lmax, wring, weighted, map, ph, fft_mode, ri);
    free(logpref); free(ri); free(ph);
    return 0;
}

int orc_sht_spin2_fast(int job, int nside, int lmax, const double* wring, double* almE, double* almB, double* mapQ,
                       double* mapU, int fft_mode, int use_mlim, int nthreads) {
    const int nring = 4 * nside - 1, npair = 2 * nside, mmax = lmax, nm = mmax + 1;
    const int synth = (job == JOB_Y || job == JOB_WY);
    const int weighted = (job == JOB_YtW || job == JOB_WY);
    const double sqrt2 = sqrt(2.0);
    enum { NVs = VL * NVV2 };
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
    cplx* phQ = (cplx*)calloc((size_t)nring * nm, sizeof(cplx));
    cplx* phU = (cplx*)calloc((size_t)nring * nm, sizeof(cplx));
    if (!phQ || !phU) return -1;
    ringinfo* ri = (ringinfo*)malloc(sizeof(ringinfo) * (nring + 1));
    for (int r = 1; r <= nring; ++r) ri[r] = ring_info(nside, r);
    if (!synth) {
        ring_stage(0, nside, lmax, wring, weighted, mapQ, phQ, fft_mode, ri);
        ring_stage(0, nside, lmax, wring, weighted, mapU, phU, fft_mode, ri);
    }
#pragma omp parallel
    {
        double* Cl = (double*)malloc(sizeof(double) * (lmax + 3));
        double* iCl = (double*)malloc(sizeof(double) * (lmax + 3));
        double* smv = (double*)malloc(sizeof(double) * (lmax + 3));
        double* Er = (double*)malloc(sizeof(double) * (lmax + 1));
        double* Ei = (double*)malloc(sizeof(double) * (lmax + 1));
        double* Br = (double*)malloc(sizeof(double) * (lmax + 1));
        double* Bi = (double*)malloc(sizeof(double) * (lmax + 1));
#pragma omp for schedule(dynamic, 1)
        for (int mi = 0; mi <= mmax; ++mi) {
            const int m = (mi & 1) ? mmax - mi / 2 : mi / 2;
            const int l0 = m > 2 ? m : 2;
            if (l0 > lmax) continue;
            for (int l = l0; l <= lmax + 1; ++l) {
                const double dl = l, dm = m;
                Cl[l] = sqrt((dl * dl - dm * dm) * (dl * dl - 4.0) / (dl * dl * (4.0 * dl * dl - 1.0)));
                iCl[l] = 1.0 / Cl[l];
                smv[l] = 2.0 * m / (dl * (dl + 1.0));
            }
            const int64_t base = mind(lmax, m);
            const double mfac = m > 0 ? sqrt2 : 1.0;
            for (int l = 0; l <= lmax; ++l) { Er[l] = Ei[l] = Br[l] = Bi[l] = 0.0; }
            if (synth)
                for (int l = l0; l <= lmax; ++l) {
                    if (m == 0) { Er[l] = almE[base + l]; Br[l] = almB[base + l]; }
                    else {
                        Er[l] = almE[base + 2 * (l - m)] * mfac; Ei[l] = almE[base + 2 * (l - m) + 1] * mfac;
                        Br[l] = almB[base + 2 * (l - m)] * mfac; Bi[l] = almB[base + 2 * (l - m) + 1] * mfac;
                    }
                }
            for (int rp0 = 1; rp0 <= npair; rp0 += NVs) {
                u4d x[NVV2], pc[NVV2], pp[NVV2], ps[NVV2], mc[NVV2], mp[NVV2], ms[NVV2];
                /* analysis: ring combinations with the parity sign folded in: [0] for l + m even, [1] for odd:
                   gW = G[odd], gX = G[1 - odd]  (gQW = GQn + sW GQs, sW = +1 even / -1 odd; gQX = GQn - sW GQs) */
                u4d QWr[2][NVV2], QWi[2][NVV2], UWr[2][NVV2], UWi[2][NVV2];
                v4d Qkr[NVV2], Qki[NVV2], Qfr[NVV2], Qfi[NVV2], Ukr[NVV2], Uki[NVV2], Ufr[NVV2], Ufi[NVV2];
                long pe[NVs], me[NVs];
                int on[NVs], any = 0;
                for (int v = 0; v < NVs; ++v) {
                    const int rp = rp0 + v, k = v / VL, j = v % VL;
                    x[k].s[j] = pc[k].s[j] = pp[k].s[j] = ps[k].s[j] = mc[k].s[j] = mp[k].s[j] = ms[k].s[j] = 0;
                    pe[v] = me[v] = 0; on[v] = 0;
                    for (int q = 0; q < 2; ++q) QWr[q][k].s[j] = QWi[q][k].s[j] = UWr[q][k].s[j] = UWi[q][k].s[j] = 0;
                    if (rp > npair) continue;
                    const ringinfo R = ri[rp];
                    if (use_mlim && m > mlim_of(lmax, 2, R.sth, R.z)) continue;
                    on[v] = 1; any = 1;
                    x[k].s[j] = R.z;
                    const double th = atan2(R.sth, R.z), c2 = cos(0.5 * th), s2 = sin(0.5 * th);
                    chain cp, cm;
                    double lg, sg;
                    sg = slam_start(2, m, c2, s2, &lg);  chain_init(&cp, sg, lg);
                    sg = slam_start(-2, m, c2, s2, &lg); chain_init(&cm, sg, lg);
                    pc[k].s[j] = cp.lc; pp[k].s[j] = cp.lp; ps[k].s[j] = cp.sf; pe[v] = cp.e;
                    mc[k].s[j] = cm.lc; mp[k].s[j] = cm.lp; ms[k].s[j] = cm.sf; me[v] = cm.e;
                    if (!synth) {
                        const cplx GQn = phQ[(size_t)(rp - 1) * nm + m], GUn = phU[(size_t)(rp - 1) * nm + m];
                        cplx GQs = 0, GUs = 0;
                        if (rp < npair) { GQs = phQ[(size_t)(4 * nside - rp - 1) * nm + m]; GUs = phU[(size_t)(4 * nside - rp - 1) * nm + m]; }
                        QWr[0][k].s[j] = creal(GQn + GQs); QWi[0][k].s[j] = cimag(GQn + GQs);
                        QWr[1][k].s[j] = creal(GQn - GQs); QWi[1][k].s[j] = cimag(GQn - GQs);
                        UWr[0][k].s[j] = creal(GUn + GUs); UWi[0][k].s[j] = cimag(GUn + GUs);
                        UWr[1][k].s[j] = creal(GUn - GUs); UWi[1][k].s[j] = cimag(GUn - GUs);
                    }
                }
                if (!any) continue;
                for (int k = 0; k < NVV2; ++k) Qkr[k] = Qki[k] = Qfr[k] = Qfi[k] = Ukr[k] = Uki[k] = Ufr[k] = Ufi[k] = vbc(0.0);
                const v4d big = vbc(RESCALE_BIG), nbig = vbc(-RESCALE_BIG), half = vbc(0.5);
                for (int lb = l0; lb <= lmax; lb += 8) {
                    const int le = lb + 8 <= lmax + 1 ? lb + 8 : lmax + 1;
                    v4l over = {0, 0, 0, 0};
                    for (int l = lb; l < le; ++l) {
                        const int odd = (l + m) & 1;
                        const v4d cl = vbc(l > l0 ? Cl[l] : 0.0), ic1 = vbc(iCl[l + 1]), sm = vbc(smv[l]);
                        if (synth) {
                            /* F^Q = -(E W + i B X), F^U = -(B W - i E X); even: W-terms keep, X-terms flip; odd: reverse */
                            const v4d er = vbc(Er[l]), ei = vbc(Ei[l]), br = vbc(Br[l]), bi = vbc(Bi[l]);
                            for (int k = 0; k < NVV2; ++k) {
                                const v4d lp2 = pc[k].v * ps[k].v, lm2 = mc[k].v * ms[k].v;
                                const v4d W = half * (lp2 + lm2), X = half * (lp2 - lm2);
                                if (!odd) {
                                    Qkr[k] -= er * W; Qki[k] -= ei * W; Qfr[k] += bi * X; Qfi[k] -= br * X;
                                    Ukr[k] -= br * W; Uki[k] -= bi * W; Ufr[k] -= ei * X; Ufi[k] += er * X;
                                } else {
                                    Qfr[k] -= er * W; Qfi[k] -= ei * W; Qkr[k] += bi * X; Qki[k] -= br * X;
                                    Ufr[k] -= br * W; Ufi[k] -= bi * W; Ukr[k] -= ei * X; Uki[k] += er * X;
                                }
                                v4d ln = ((x[k].v + sm) * pc[k].v - cl * pp[k].v) * ic1;
                                pp[k].v = pc[k].v; pc[k].v = ln;
                                over |= (ln > big) | (ln < nbig);
                                ln = ((x[k].v - sm) * mc[k].v - cl * mp[k].v) * ic1;
                                mp[k].v = mc[k].v; mc[k].v = ln;
                                over |= (ln > big) | (ln < nbig);
                            }
                        } else {
                            /* E += -W gQW - i X gUX ; B += -W gUW + i X gQX */
                            v4d ser = vbc(0.0), sei = vbc(0.0), sbr = vbc(0.0), sbi = vbc(0.0);
                            for (int k = 0; k < NVV2; ++k) {
                                const v4d lp2 = pc[k].v * ps[k].v, lm2 = mc[k].v * ms[k].v;
                                const v4d W = half * (lp2 + lm2), X = half * (lp2 - lm2);
                                ser += X * UWi[1 - odd][k].v - W * QWr[odd][k].v;
                                sei -= W * QWi[odd][k].v + X * UWr[1 - odd][k].v;
                                sbr -= W * UWr[odd][k].v + X * QWi[1 - odd][k].v;
                                sbi += X * QWr[1 - odd][k].v - W * UWi[odd][k].v;
                                v4d ln = ((x[k].v + sm) * pc[k].v - cl * pp[k].v) * ic1;
                                pp[k].v = pc[k].v; pc[k].v = ln;
                                over |= (ln > big) | (ln < nbig);
                                ln = ((x[k].v - sm) * mc[k].v - cl * mp[k].v) * ic1;
                                mp[k].v = mc[k].v; mc[k].v = ln;
                                over |= (ln > big) | (ln < nbig);
                            }
                            Er[l] += vsum(ser); Ei[l] += vsum(sei); Br[l] += vsum(sbr); Bi[l] += vsum(sbi);
                        }
                    }
                    u4l ov; ov.v = over;
                    if (ov.s[0] | ov.s[1] | ov.s[2] | ov.s[3])
                        for (int v = 0; v < NVs; ++v) {
                            const int k = v / VL, j = v % VL;
                            if (fabs(pc[k].s[j]) > RESCALE_BIG) {
                                pc[k].s[j] *= RESCALE_INV; pp[k].s[j] *= RESCALE_INV; pe[v] += 300;
                                ps[k].s[j] = (pe[v] < -900) ? 0.0 : ldexp(1.0, (int)pe[v]);
                            }
                            if (fabs(mc[k].s[j]) > RESCALE_BIG) {
                                mc[k].s[j] *= RESCALE_INV; mp[k].s[j] *= RESCALE_INV; me[v] += 300;
                                ms[k].s[j] = (me[v] < -900) ? 0.0 : ldexp(1.0, (int)me[v]);
                            }
                        }
                }
                if (synth)
                    for (int v = 0; v < NVs; ++v) {
                        const int rp = rp0 + v, k = v / VL, j = v % VL;
                        if (!on[v]) continue;
                        u4d a, b, c, d, e2, f, g, h;
                        a.v = Qkr[k]; b.v = Qki[k]; c.v = Qfr[k]; d.v = Qfi[k]; e2.v = Ukr[k]; f.v = Uki[k]; g.v = Ufr[k]; h.v = Ufi[k];
                        phQ[(size_t)(rp - 1) * nm + m] = (a.s[j] + c.s[j]) + I * (b.s[j] + d.s[j]);
                        phU[(size_t)(rp - 1) * nm + m] = (e2.s[j] + g.s[j]) + I * (f.s[j] + h.s[j]);
                        if (rp < npair) {
                            phQ[(size_t)(4 * nside - rp - 1) * nm + m] = (a.s[j] - c.s[j]) + I * (b.s[j] - d.s[j]);
                            phU[(size_t)(4 * nside - rp - 1) * nm + m] = (e2.s[j] - g.s[j]) + I * (f.s[j] - h.s[j]);
                        }
                    }
            }
            if (!synth)
                for (int l = m; l <= lmax; ++l) {
                    if (m == 0) { almE[base + l] = Er[l]; almB[base + l] = Br[l]; }
                    else {
                        almE[base + 2 * (l - m)] = Er[l] * mfac; almE[base + 2 * (l - m) + 1] = Ei[l] * mfac;
                        almB[base + 2 * (l - m)] = Br[l] * mfac; almB[base + 2 * (l - m) + 1] = Bi[l] * mfac;
                    }
                }
        }
        free(Cl); free(iCl); free(smv); free(Er); free(Ei); free(Br); free(Bi);
    }
    if (synth) {
        ring_stage(1, nside, lmax, wring, weighted, mapQ, phQ, fft_mode, ri);
        ring_stage(1, nside, lmax, wring, weighted, mapU, phU, fft_mode, ri);
    }
    free(ri); free(phQ); free(phU);
    return 0;
}
