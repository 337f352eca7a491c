// Device bodies of the SHT kernels, written so that the same source also compiles as plain C++ (CMDR_HD empty)
// for the single-thread host emulation used by the CPU-only index/math tests (tests/host_emul).
//
// The SHT is libsharp2's decomposition re-designed for CDNA4 wavefronts (NOT a port; libsharp2 is not in this
// tree): per-ring FFTs <-> phases F_m(ring) <-> per-m Legendre recursions.
//   * Legendre: one wavefront = one m x (64*R) ring pairs; lane = ring pair, l runs sequentially and
//     wave-uniformly so a_lm / recursion coefficients come in through the scalar unit.  The three-term
//     recursion is renormalised (lambda = cnorm * mu, mu_l = alpha_l x mu_{l-1} - mu_{l-2}) to 2 fp64 ops per l,
//     and starts from tabulated seeds (ls, mu_ls, mu_ls-1) instead of a scaled run-up from l = m.
//   * Rings: one workgroup = one north/south ring pair packed into a single complex FFT held in LDS
//     (power-of-two rings directly, polar-cap rings through Bluestein), optionally fusing the per-pixel N^-1
//     so the map never leaves the CU between synthesis and adjoint.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define CMDR_HD __host__ __device__ __forceinline__
#else
#define CMDR_HD inline
#endif

namespace cmdr {

constexpr int kAdjL_ = 8;  // l values per transpose-reduce group of the adjoint Legendre kernel

struct cd {
    double x, y;
};
CMDR_HD cd cmul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
CMDR_HD cd cadd(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
CMDR_HD cd csub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
CMDR_HD cd cconj(cd a) { return {a.x, -a.y}; }

CMDR_HD int64_t d_moffp(int lmax, int m) { return (int64_t)m * (lmax + 2) - (int64_t)m * (m - 1) / 2; }

struct LegArgs {
    int lmax;
    int npair_pad;
    int R;                 // ring pairs per lane of the kernel these args are passed to
    const double* x;       // [npair_pad]
    const int* ls;         // [(lmax+1) * npair_pad]
    const double* seedc;   // mu_{ls}
    const double* seedp;   // mu_{ls-1}
    const double* alpha;   // [ntrip]
};

// ---------------------------------------------------------------------------------------------------------
// Legendre synthesis, one lane, NB maps at once: F^(k)_N/S(m, pair) = sum_l a~^(k)_lm mu_l(x_pair).
// The recursion (2 fp64 ops per l) is shared by the NB maps, each map adds 2 FMAs per l.
//   ast : coefficient stream, complex, padded-triangle layout, maps interleaved: ast[((t*nbs)+k)*2 + {re,im}],
//         already multiplied by cnorm etc.;  k0 = first map of this launch slice
//   ph  : phase arrays [map][(lmax+1)][npair_pad][4] = (N.re, N.im, S.re, S.im)
template <int R, int NB>
CMDR_HD void leg_synth_lane(const LegArgs& A, const double* __restrict__ ast, int nbs, int k0,
                            double* __restrict__ ph, int64_t ph_stride, int m, int chunk, int lw, int lAend,
                            int lane) {
    const int lmax = A.lmax;
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ as = ast + 2 * ((int64_t)nbs * (mo - m) + k0);
    const int64_t ls2 = 2 * (int64_t)nbs;  // stride between consecutive l
    double x[R], mc[R], mp[R], sc[R], sp[R];
    double Er[R][NB], Ei[R][NB], Or[R][NB], Oi[R][NB];
    int ls[R];
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x[r] = A.x[p];
        ls[r] = A.ls[idx];
        sc[r] = A.seedc[idx];
        sp[r] = A.seedp[idx];
        mc[r] = mp[r] = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) Er[r][k] = Ei[r][k] = Or[r][k] = Oi[r][k] = 0.0;
    }
    int l = lw;
    // Phase A: lanes switch on at their own ls
    for (; l < lAend && l <= lmax; l += 2) {
        const double* __restrict__ c0 = as + ls2 * l;
        const double* __restrict__ c1 = c0 + ls2;
        const double al1 = al[l + 1], al2 = al[l + 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (ls[r] == l) { mc[r] = sc[r]; mp[r] = sp[r]; }
#pragma unroll
            for (int k = 0; k < NB; ++k) { Er[r][k] += mc[r] * c0[2 * k]; Ei[r][k] += mc[r] * c0[2 * k + 1]; }
            double t = al1 * x[r] * mc[r] - mp[r];
            mp[r] = mc[r];
            mc[r] = t;
            if (ls[r] == l + 1) { mc[r] = sc[r]; mp[r] = sp[r]; }
#pragma unroll
            for (int k = 0; k < NB; ++k) { Or[r][k] += mc[r] * c1[2 * k]; Oi[r][k] += mc[r] * c1[2 * k + 1]; }
            t = al2 * x[r] * mc[r] - mp[r];
            mp[r] = mc[r];
            mc[r] = t;
        }
    }
    // Phase B: every started lane is running; pure recursion + accumulate.  The wave-uniform coefficients come in
    // through scalar loads.  Scalar loads return out of order (only lgkmcnt(0) exists), so a wave cannot pipeline
    // its own loads and their ~500-800 cycle latency (scalar-cache misses to L2; measured) must be covered by the
    // other waves of the SIMD: each trip therefore consumes FOUR l (one batch of loads, one wait) when the SGPR
    // budget allows, which makes the VALU block between two waits long enough for 4 waves/SIMD to hide it.
    constexpr bool kQuad = (2 + 4 * NB) * 4 <= 76;  // SGPRs for 4 l of (alpha, NB complex)
    if (kQuad) {
        for (; l + 2 <= lmax; l += 4) {
            const double* __restrict__ c0 = as + ls2 * l;
            const double* __restrict__ c1 = c0 + ls2;
            const double* __restrict__ c2 = c1 + ls2;
            const double* __restrict__ c3 = c2 + ls2;
            const double al1 = al[l + 1], al2 = al[l + 2], al3 = al[l + 3], al4 = al[l + 4];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < NB; ++k) { Er[r][k] += mc[r] * c0[2 * k]; Ei[r][k] += mc[r] * c0[2 * k + 1]; }
                double t = al1 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = t;
#pragma unroll
                for (int k = 0; k < NB; ++k) { Or[r][k] += mc[r] * c1[2 * k]; Oi[r][k] += mc[r] * c1[2 * k + 1]; }
                t = al2 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = t;
#pragma unroll
                for (int k = 0; k < NB; ++k) { Er[r][k] += mc[r] * c2[2 * k]; Ei[r][k] += mc[r] * c2[2 * k + 1]; }
                t = al3 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = t;
#pragma unroll
                for (int k = 0; k < NB; ++k) { Or[r][k] += mc[r] * c3[2 * k]; Oi[r][k] += mc[r] * c3[2 * k + 1]; }
                t = al4 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = t;
            }
        }
    }
    for (; l <= lmax; l += 2) {
        const double* __restrict__ c0 = as + ls2 * l;
        const double* __restrict__ c1 = c0 + ls2;
        const double al1 = al[l + 1], al2 = al[l + 2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < NB; ++k) { Er[r][k] += mc[r] * c0[2 * k]; Ei[r][k] += mc[r] * c0[2 * k + 1]; }
            double t = al1 * x[r] * mc[r] - mp[r];
            mp[r] = mc[r];
            mc[r] = t;
#pragma unroll
            for (int k = 0; k < NB; ++k) { Or[r][k] += mc[r] * c1[2 * k]; Oi[r][k] += mc[r] * c1[2 * k + 1]; }
            t = al2 * x[r] * mc[r] - mp[r];
            mp[r] = mc[r];
            mc[r] = t;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            double* o = ph + (k0 + k) * ph_stride + ((int64_t)m * A.npair_pad + p) * 4;
            o[0] = Er[r][k] + Or[r][k];
            o[1] = Ei[r][k] + Oi[r][k];
            o[2] = Er[r][k] - Or[r][k];
            o[3] = Ei[r][k] - Oi[r][k];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Legendre adjoint: per-lane state, NB maps sharing one recursion, groups of kAdjL_ (= 8) l values.
template <int R, int NB>
struct AdjLane {
    double x[R], mc[R], mp[R], sc[R], sp[R];
    double Ger[R][NB], Gei[R][NB], Gor[R][NB], Goi[R][NB];
    int ls[R];
};

// SQUARE: accumulate mu^2 * G (used for the harmonic-space noise diagonal, comm_N_mod.f90:127-197)
template <int R, int NB, bool SQUARE>
CMDR_HD void leg_adj_load(const LegArgs& A, const double* __restrict__ ph, int64_t ph_stride, int k0, int m,
                          int chunk, int lane, AdjLane<R, NB>& S) {
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        S.x[r] = A.x[p];
        S.ls[r] = A.ls[idx];
        S.sc[r] = A.seedc[idx];
        S.sp[r] = A.seedp[idx];
        S.mc[r] = S.mp[r] = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const double* g = ph + (k0 + k) * ph_stride + idx * 4;
            const double nr = g[0], ni = g[1], sr = g[2], si = g[3];
            S.Ger[r][k] = nr + sr;
            S.Gei[r][k] = ni + si;
            if (SQUARE) { S.Gor[r][k] = nr + sr; S.Goi[r][k] = ni + si; }
            else        { S.Gor[r][k] = nr - sr; S.Goi[r][k] = ni - si; }
        }
    }
}

// Advance the recursion over l0 .. l0+7 and keep the (possibly squared) mu values: w[j][r]
template <int R, int NB, bool SQUARE, bool INJECT>
CMDR_HD void leg_adj_mu_group(const double* __restrict__ al, int l0, AdjLane<R, NB>& S, double (*w)[R]) {
#pragma unroll
    for (int j = 0; j < kAdjL_; ++j) {
        const int l = l0 + j;
        const double al1 = al[l + 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (INJECT) if (S.ls[r] == l) { S.mc[r] = S.sc[r]; S.mp[r] = S.sp[r]; }
            w[j][r] = SQUARE ? S.mc[r] * S.mc[r] : S.mc[r];
            const double t = al1 * S.x[r] * S.mc[r] - S.mp[r];
            S.mp[r] = S.mc[r];
            S.mc[r] = t;
        }
    }
}

// v[2*j], v[2*j+1] = (re, im) partial sums of this lane for l = l0 + j (j < 8), map k
template <int R, int NB>
CMDR_HD void leg_adj_products(const AdjLane<R, NB>& S, const double (*w)[R], int k, double* v) {
#pragma unroll
    for (int j = 0; j < kAdjL_; j += 2) {
        double tr = 0.0, ti = 0.0, ur = 0.0, ui = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            tr += w[j][r] * S.Ger[r][k];
            ti += w[j][r] * S.Gei[r][k];
            ur += w[j + 1][r] * S.Gor[r][k];
            ui += w[j + 1][r] * S.Goi[r][k];
        }
        v[2 * j] = tr;
        v[2 * j + 1] = ti;
        v[2 * j + 2] = ur;
        v[2 * j + 3] = ui;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Ring FFT pieces.  buf = LDS array of M complex; tw = exp(+2 pi i k / Mmax), k < Mmax/2.
struct FftCtx {
    int tid, nthr;
};

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define CMDR_BLOCK_SYNC() __syncthreads()
#else
#define CMDR_BLOCK_SYNC() ((void)0)
#endif

// natural order in -> bit-reversed order out, kernel exp(+2 pi i jk/M)
CMDR_HD void fft_dif_plus(cd* buf, int log2M, const cd* __restrict__ tw, int log2Mmax, FftCtx c) {
    const int M = 1 << log2M;
    for (int s = log2M; s >= 1; --s) {
        const int half = 1 << (s - 1);
        const int tws = log2Mmax - s;  // twiddle stride = Mmax/len
        for (int b = c.tid; b < (M >> 1); b += c.nthr) {
            const int pos = b & (half - 1);
            const int i0 = ((b >> (s - 1)) << s) + pos;
            const int i1 = i0 + half;
            const cd u = buf[i0], v = buf[i1];
            buf[i0] = cadd(u, v);
            buf[i1] = cmul(csub(u, v), tw[pos << tws]);
        }
        CMDR_BLOCK_SYNC();
    }
}

// bit-reversed order in -> natural order out, kernel exp(+2 pi i jk/M)
CMDR_HD void fft_dit_plus(cd* buf, int log2M, const cd* __restrict__ tw, int log2Mmax, FftCtx c) {
    const int M = 1 << log2M;
    for (int s = 1; s <= log2M; ++s) {
        const int half = 1 << (s - 1);
        const int tws = log2Mmax - s;
        for (int b = c.tid; b < (M >> 1); b += c.nthr) {
            const int pos = b & (half - 1);
            const int i0 = ((b >> (s - 1)) << s) + pos;
            const int i1 = i0 + half;
            const cd u = buf[i0], v = cmul(buf[i1], tw[pos << tws]);
            buf[i0] = cadd(u, v);
            buf[i1] = csub(u, v);
        }
        CMDR_BLOCK_SYNC();
    }
}

CMDR_HD int d_bitrev(int v, int bits) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    return (int)(__brev((unsigned)v) >> (32 - bits));
#else
    int r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
#endif
}

struct RingDev {  // device mirror of RingPairDesc
    int nphi, log2M, bluestein, mmax_eff;
    long long startN, startS;
    double phi0, wgt;
    long long chirp_off;
    int ring, pad;
};

// exp(i * m * phi0) for HEALPix: phi0 = pi/(4 i) (cap), pi/(4 N) or 0 (belt) -> exact argument reduction on
// integers before the libm call: m*phi0 = 2 pi (m mod 8q) / (8 q).
CMDR_HD cd phase_mphi0(int m, int q /* 8q = period; q=0: phi0=0 */) {
    if (q == 0) return {1.0, 0.0};
    const int per = 8 * q;
    const int k = m % per;
    const double ang = 6.283185307179586476925286766559 * (double)k / (double)per;
    double s, c;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    sincos(ang, &s, &c);
#else
    s = __builtin_sin(ang);
    c = __builtin_cos(ang);
#endif
    return {c, s};
}

// Build the packed spectrum Z_j = X^N_j + i X^S_j (j < n) of a ring pair from its phases, store at buf[pos(j)].
//   X[m mod n] += G_m ; X[(-m) mod n] += conj(G_m) (m>0) ; G_m = kappa_m F_m e^{i m phi0}; kappa = 1 (m=0) (sqrt2 is
//   already folded into the coefficient stream).
// Gather form (slot j sums its aliases) so no atomics are needed.  For Bluestein the slot is pre-multiplied by
// the chirp and conjugated (see ring_synth_finish).
CMDR_HD cd ring_gather_slot(const double* __restrict__ ph, int64_t npair_pad, int pair, int n, int mmax, int q,
                            int j) {
    cd z = {0.0, 0.0};
    for (int m = j; m <= mmax; m += n) {  // direct aliases
        const double* f = ph + ((int64_t)m * npair_pad + pair) * 4;
        const cd e = phase_mphi0(m, q);
        const cd gn = cmul({f[0], f[1]}, e), gs = cmul({f[2], f[3]}, e);
        // Z += gn + i gs
        z.x += gn.x - gs.y;
        z.y += gn.y + gs.x;
    }
    for (int m = (j == 0 ? n : n - j); m <= mmax; m += n) {  // conjugate aliases (m > 0)
        const double* f = ph + ((int64_t)m * npair_pad + pair) * 4;
        const cd e = phase_mphi0(m, q);
        const cd gn = cconj(cmul({f[0], f[1]}, e)), gs = cconj(cmul({f[2], f[3]}, e));
        z.x += gn.x - gs.y;
        z.y += gn.y + gs.x;
    }
    return z;
}

// Full inverse (synthesis) ring transform in LDS: on return buf[k], k<n holds y^N_k + i y^S_k (natural order).
CMDR_HD void ring_synth_lds(cd* buf, const RingDev& d, const double* __restrict__ ph, int64_t npair_pad, int pair,
                            const cd* __restrict__ tw, int log2Mmax, const cd* __restrict__ chirp, FftCtx c) {
    const int n = d.nphi, M = 1 << d.log2M;
    const int q = d.phi0 == 0.0 ? 0 : n / 4;  // phi0 = pi/(4i) with n = 4i (cap) or pi/(4N) with n = 4N (belt)
    if (!d.bluestein) {
        for (int j = c.tid; j < n; j += c.nthr)
            buf[d_bitrev(j, d.log2M)] = ring_gather_slot(ph, npair_pad, pair, n, d.mmax_eff, q, j);
        CMDR_BLOCK_SYNC();
        fft_dit_plus(buf, d.log2M, tw, log2Mmax, c);
    } else {
        const cd* w = chirp + d.chirp_off;   // w_j, j<n
        const cd* chat = w + n;              // bit-reversed FFT_M^- of the conj chirp
        // a_j = Z_j w_j ; we need F^-(a) = conj(F^+(conj a))
        for (int j = c.tid; j < M; j += c.nthr) {
            cd v = {0.0, 0.0};
            if (j < n) v = cconj(cmul(ring_gather_slot(ph, npair_pad, pair, n, d.mmax_eff, q, j), w[j]));
            buf[j] = v;
        }
        CMDR_BLOCK_SYNC();
        fft_dif_plus(buf, d.log2M, tw, log2Mmax, c);
        for (int p = c.tid; p < M; p += c.nthr) buf[p] = cmul(cconj(buf[p]), chat[p]);
        CMDR_BLOCK_SYNC();
        fft_dit_plus(buf, d.log2M, tw, log2Mmax, c);
        const double inv = 1.0 / (double)M;
        for (int k = c.tid; k < n; k += c.nthr) {
            const cd v = cmul(buf[k], w[k]);
            buf[k] = {v.x * inv, v.y * inv};
        }
        CMDR_BLOCK_SYNC();
    }
}

// Forward (analysis) ring transform in LDS.  On entry buf[k], k<n holds z_k = y^N_k + i y^S_k (natural order).
// On return the spectrum Z_j = sum_k z_k e^{-2 pi i jk/n} is available through ring_spec_at().
CMDR_HD void ring_anal_lds(cd* buf, const RingDev& d, const cd* __restrict__ tw, int log2Mmax,
                           const cd* __restrict__ chirp, FftCtx c) {
    const int n = d.nphi, M = 1 << d.log2M;
    if (!d.bluestein) {
        // Z = conj(F^+(conj z)); DIF leaves it bit-reversed
        for (int k = c.tid; k < n; k += c.nthr) buf[k] = cconj(buf[k]);
        CMDR_BLOCK_SYNC();
        fft_dif_plus(buf, d.log2M, tw, log2Mmax, c);
    } else {
        // conj(Z)_j = sum_k conj(z_k) e^{+...} = w_j sum_k (conj(z_k) w_k) conj(w_{j-k})  (same chirp machinery)
        const cd* w = chirp + d.chirp_off;
        const cd* chat = w + n;
        for (int k = c.tid; k < M; k += c.nthr) {
            cd v = {0.0, 0.0};
            if (k < n) v = cconj(cmul(cconj(buf[k]), w[k]));   // conj(a_k), a_k = conj(z_k) w_k
            buf[k] = v;
        }
        CMDR_BLOCK_SYNC();
        fft_dif_plus(buf, d.log2M, tw, log2Mmax, c);
        for (int p = c.tid; p < M; p += c.nthr) buf[p] = cmul(cconj(buf[p]), chat[p]);
        CMDR_BLOCK_SYNC();
        fft_dit_plus(buf, d.log2M, tw, log2Mmax, c);
        const double inv = 1.0 / (double)M;
        for (int j = c.tid; j < n; j += c.nthr) {
            const cd v = cmul(buf[j], w[j]);      // conj(Z_j) * M
            buf[j] = {v.x * inv, -v.y * inv};     // Z_j, natural order
        }
        CMDR_BLOCK_SYNC();
    }
}

CMDR_HD cd ring_spec_at(const cd* buf, const RingDev& d, int j) {
    if (!d.bluestein) return cconj(buf[d_bitrev(j, d.log2M)]);
    return buf[j];
}

// Extract G^N_m, G^S_m (m <= mmax_eff) from the packed spectrum and store them as phases for the adjoint
// Legendre stage: X^N_j = (Z_j + conj Z_{n-j})/2, X^S_j = (Z_j - conj Z_{n-j})/(2i); G_m = X[m mod n] e^{-i m phi0}.
CMDR_HD void ring_store_phases(const cd* buf, const RingDev& d, double* __restrict__ ph, int64_t npair_pad,
                               int pair, FftCtx c) {
    const int n = d.nphi;
    const int q = d.phi0 == 0.0 ? 0 : n / 4;
    for (int m = c.tid; m <= d.mmax_eff; m += c.nthr) {
        const int j = m % n;
        const cd a = ring_spec_at(buf, d, j), b = cconj(ring_spec_at(buf, d, j == 0 ? 0 : n - j));
        const cd xn = {0.5 * (a.x + b.x), 0.5 * (a.y + b.y)};
        const cd dm = {0.5 * (a.x - b.x), 0.5 * (a.y - b.y)};
        const cd xs = {dm.y, -dm.x};  // dm / i
        const cd e = cconj(phase_mphi0(m, q));
        const cd gn = cmul(xn, e), gs = cmul(xs, e);
        double* o = ph + ((int64_t)m * npair_pad + pair) * 4;
        o[0] = gn.x;
        o[1] = gn.y;
        o[2] = gs.x;
        o[3] = gs.y;
    }
}

// ---------------------------------------------------------------------------------------------------------
// a_lm layout conversion elements (one (l, m) each).
// Commander real-packed index of (l, +m) for a full (P=1) layout: comm_map_mod.f90:228-261, :1213-1246.
CMDR_HD int64_t d_packed_index(int lmax, int l, int m) {
    if (m == 0) return l;
    return 2 * ((int64_t)m * (lmax + 1) - (int64_t)m * (m - 1) / 2) - (lmax + 1) + 2 * (l - m);
}

// packed a_lm -> padded-triangle complex stream entry, times cnorm * kappa_m; kappa = 1/sqrt2 for m > 0
// (the ring stage builds X[m] += G, X[-m] += conj G), 1 for m = 0.  l = lmax+1 is the zero pad entry.
CMDR_HD void alm_to_stream_elem(const double* __restrict__ a, double* __restrict__ ast, int nbs, int k,
                                const double* __restrict__ cnorm, int lmax, int m, int l) {
    const int64_t t = d_moffp(lmax, m) + (l - m);
    double* __restrict__ o = ast + 2 * (t * nbs + k) - 2 * t;  // so that o[2t] is this map's slot
    double re = 0.0, im = 0.0;
    if (l <= lmax) {
        const double cn = cnorm[t];
        const int64_t i = d_packed_index(lmax, l, m);
        if (m == 0) {
            re = a[i] * cn;
        } else {
            const double f = cn * 0.70710678118654752440;
            re = a[i] * f;
            im = a[i + 1] * f;
        }
    }
    o[2 * t] = re;
    o[2 * t + 1] = im;
}

// adjoint partial columns -> packed a_lm: kappa'_m * cnorm * sum_chunks part ; kappa' = sqrt2 for m > 0.
CMDR_HD void part_to_alm_elem(const double* __restrict__ p, int64_t part_chunk_stride, int nchunk,
                              double* __restrict__ a, const double* __restrict__ cnorm, int lmax, int m, int l) {
    const int64_t t = d_moffp(lmax, m) + (l - m);
    double re = 0.0, im = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        re += p[c * part_chunk_stride + 2 * t];
        im += p[c * part_chunk_stride + 2 * t + 1];
    }
    const double cn = cnorm[t];
    const int64_t i = d_packed_index(lmax, l, m);
    if (m == 0) {
        a[i] = re * cn;
    } else {
        const double f = cn * 1.41421356237309504880;
        a[i] = re * f;
        a[i + 1] = im * f;
    }
}

}  // namespace cmdr
