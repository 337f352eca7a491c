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
#include <cmath>
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
// Commander real-packed index of (l, +m) for a full (P=1) layout: comm_map_mod.f90:228-261, :1213-1246.
CMDR_HD int64_t d_packed_index(int lmax, int l, int m) {
    if (m == 0) return l;
    return 2 * ((int64_t)m * (lmax + 1) - (int64_t)m * (m - 1) / 2) - (lmax + 1) + 2 * (l - m);
}

// Phase arrays  F_m(ring pair) = (N.re, N.im, S.re, S.im), layout [map][pair][m][4]: the m of one ring pair are contiguous,
// so the ring stage -- whose time goes into loading and storing exactly these entries -- streams 32 (m_max + 1) bytes per
// pair fully coalesced.  The Legendre kernels (lane = pair) touch them once per task with a stride of one row per lane;
// they are compute-bound and hide that.  prow = rows per pair (lmax + 1; 2 m_max + 1 rows in the Toeplitz setup array).
// (A tiled variant [pair / 16][m][pair % 16] -- 512-byte runs for the Legendre lanes, a 1 MB region per 16 pairs for the
// ring stage -- measured the same matvec time: ring +0.17 ms, Legendre -0.1 ms; the plain form is kept.)
CMDR_HD int64_t d_phidx(int64_t prow, int64_t pair, int m) { return (pair * prow + m) * 4; }

struct LegArgs {
    int lmax;
    int npair_pad;
    int R;                 // ring pairs per lane of the kernel these args are passed to
    const double* x;       // [npair_pad]
    const int* ls;         // [(lmax+1) * npair_pad]
    const double* seedc;   // mu_{ls}
    const double* seedp;   // mu_{ls-1}
    const double* alpha;   // [ntrip]
    int wg;                // synthesis task list grouped by m: the 4 tasks of a workgroup are 4 chunks of one m
    int uni;               // every 64-pair lane block starts at one l == m (mod 32) (LegendreTables::uniform_start)
};

// ---------------------------------------------------------------------------------------------------------
// Legendre synthesis, one lane, NB maps at once: F^(k)_N/S(m, pair) = sum_l a~^(k)_lm mu_l(x_pair).
// The recursion (2 fp64 ops per l) is shared by the NB maps, each map adds 2 FMAs per l.
//   ast : coefficient stream, complex, padded-triangle layout, maps interleaved: ast[((t*nbs)+k)*2 + {re,im}],
//         already multiplied by cnorm etc.;  k0 = first map of this launch slice
//   ph  : phase arrays [map][npair_pad][lmax+1][4] = (N.re, N.im, S.re, S.im)  (d_phidx)
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
            double* o = ph + (k0 + k) * ph_stride + d_phidx(A.lmax + 1, p, m);
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
            const double* g = ph + (k0 + k) * ph_stride + d_phidx(A.lmax + 1, p, m);
            const double nr = g[0], ni = g[1], sr = g[2], si = g[3];
            S.Ger[r][k] = nr + sr;
            S.Gei[r][k] = ni + si;
            if (SQUARE) { S.Gor[r][k] = nr + sr; S.Goi[r][k] = ni + si; }
            else        { S.Gor[r][k] = nr - sr; S.Goi[r][k] = ni - si; }
        }
    }
}

// Advance the recursion over l0 .. l0+7 and keep the (possibly squared) mu values: w[j][r]
// a8[j] = alpha_{l0 + j + 1}
template <int R, int NB, bool SQUARE, bool INJECT>
CMDR_HD void leg_adj_mu_group_v(const double (&a8)[kAdjL_], int l0, AdjLane<R, NB>& S, double (*w)[R]) {
#pragma unroll
    for (int j = 0; j < kAdjL_; ++j) {
        const int l = l0 + j;
        const double al1 = a8[j];
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
template <int R, int NB, bool SQUARE, bool INJECT>
CMDR_HD void leg_adj_mu_group(const double* __restrict__ al, int l0, AdjLane<R, NB>& S, double (*w)[R]) {
    double a8[kAdjL_];
#pragma unroll
    for (int j = 0; j < kAdjL_; ++j) a8[j] = al[l0 + j + 1];
    leg_adj_mu_group_v<R, NB, SQUARE, INJECT>(a8, l0, S, w);
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
// Spin-2 Legendre stage: (E,B) stream -> phases of the Q and U maps (and its transpose).  Two spin-weighted
// chains mu+ / mu- per ring pair; W = mu+ + mu-, X = mu+ - mu- (the 1/2 and the minus signs of
//   F^Q = -sum_l (E W + i B X),   F^U = -sum_l (B W - i E X)
// are folded into the stream: it carries E' = -E cnorm kappa / 2, B' likewise).  W has parity (-1)^{l+m} under
// theta -> pi - theta, X the opposite, so each l feeds a "keep" and a "flip" accumulator: F_N = k + f, F_S = k - f.
struct Leg2Args {
    int lmax;
    int npair_pad;
    int R;
    const double* x;       // [npair_pad]
    const int* ls;         // [(lmax+1) * npair_pad]
    const double* seed;    // [(lmax+1) * npair_pad * 4]
    const double* alpha;   // [ntrip]
    const double* beta;    // [ntrip]
    const double* abs_;    // [2 ntrip] (alpha, beta)_l times (-1)^(l - 1 - l0), interleaved: the sign-alternated form of the
                           // recursion used by k_leg2_adj_px (nu_{l+1} = nu_{l-1} + (alphaS x +- betaS) nu_l, no subtraction)
};

template <int R>
struct Leg2State {
    double x[R], pc[R], pp[R], mc[R], mp[R];   // mu+ (current, previous), mu- (current, previous)
    const double* sdp;                         // seeds of ring pair r at sdp + 256 r: fetched when the pair starts
    int ls[R];                                 // (once per ring pair and m), not carried in registers
    CMDR_HD double sd(int r, int k) const { return sdp[r * 256 + k]; }
};

template <int R>
CMDR_HD void leg2_load_state(const Leg2Args& A, int m, int chunk, int lane, Leg2State<R>& S) {
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        S.x[r] = A.x[p];
        S.ls[r] = A.ls[idx];
        if (r == 0) S.sdp = A.seed + idx * 4;
        S.pc[r] = S.pp[r] = S.mc[r] = S.mp[r] = 0.0;
    }
}

template <int R, bool INJECT>
CMDR_HD void leg2_advance(Leg2State<R>& S, int r, int l, double al, double be) {
    // on entry the state is at l; on exit at l+1 (al, be are alpha_{l+1}, beta_{l+1})
    const double tp = al * S.x[r] + be, tm = al * S.x[r] - be;
    double n = tp * S.pc[r] - S.pp[r];
    S.pp[r] = S.pc[r];
    S.pc[r] = n;
    n = tm * S.mc[r] - S.mp[r];
    S.mp[r] = S.mc[r];
    S.mc[r] = n;
    if (INJECT) if (S.ls[r] == l + 1) { S.pc[r] = S.sd(r, 0); S.pp[r] = S.sd(r, 1); S.mc[r] = S.sd(r, 2); S.mp[r] = S.sd(r, 3); }
}

// st : (E,B) stream of this polarisation pair set: st[((t * npol) + ip) * 4 + {E'r, E'i, B'r, B'i}]
// ph : phase arrays; map kq = Q, kq + 1 = U
template <int R>
CMDR_HD void leg2_synth_lane(const Leg2Args& A, const double* __restrict__ st, int npol, int ip,
                             double* __restrict__ ph, int64_t ph_stride, int kq, int m, int chunk, int lw,
                             int lAend, int lane) {
    const int lmax = A.lmax;
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    const double* __restrict__ as = st + 4 * ((int64_t)npol * (mo - m) + ip);
    const int64_t ls4 = 4 * (int64_t)npol;
    Leg2State<R> S;
    leg2_load_state<R>(A, m, chunk, lane, S);
    // accumulators: [0] = Q keep, [1] = Q flip, [2] = U keep, [3] = U flip ; (re, im)
    double ar[R][4], ai[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) ar[r][k] = ai[r][k] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (S.ls[r] == lw) { S.pc[r] = S.sd(r, 0); S.pp[r] = S.sd(r, 1); S.mc[r] = S.sd(r, 2); S.mp[r] = S.sd(r, 3); }
    for (int l = lw; l <= lmax; l += 2) {
        const double* __restrict__ c0 = as + ls4 * l;
        const double* __restrict__ c1 = c0 + ls4;
        const double e0r = c0[0], e0i = c0[1], b0r = c0[2], b0i = c0[3];
        const double e1r = c1[0], e1i = c1[1], b1r = c1[2], b1i = c1[3];
        const double al1 = al[l + 1], be1 = be[l + 1], al2 = al[l + 2], be2 = be[l + 2];
        const bool inj = l < lAend;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double W = S.pc[r] + S.mc[r], X = S.pc[r] - S.mc[r];
            // first of the pair: W-terms keep, X-terms flip
            ar[r][0] += e0r * W;  ai[r][0] += e0i * W;      // Q keep  += E' W
            ar[r][1] -= b0i * X;  ai[r][1] += b0r * X;      // Q flip  += i B' X
            ar[r][2] += b0r * W;  ai[r][2] += b0i * W;      // U keep  += B' W
            ar[r][3] += e0i * X;  ai[r][3] -= e0r * X;      // U flip  += -i E' X
            if (inj) leg2_advance<R, true>(S, r, l, al1, be1); else leg2_advance<R, false>(S, r, l, al1, be1);
            W = S.pc[r] + S.mc[r];
            X = S.pc[r] - S.mc[r];
            // second of the pair: W-terms flip, X-terms keep
            ar[r][1] += e1r * W;  ai[r][1] += e1i * W;
            ar[r][0] -= b1i * X;  ai[r][0] += b1r * X;
            ar[r][3] += b1r * W;  ai[r][3] += b1i * W;
            ar[r][2] += e1i * X;  ai[r][2] -= e1r * X;
            if (inj) leg2_advance<R, true>(S, r, l + 1, al2, be2); else leg2_advance<R, false>(S, r, l + 1, al2, be2);
        }
    }
    const int l0 = m > 2 ? m : 2;
    const bool swap = ((l0 + m) & 1) != 0;   // first-of-pair has odd l+m: keep <-> flip
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double kr = swap ? ar[r][2 * q + 1] : ar[r][2 * q], ki = swap ? ai[r][2 * q + 1] : ai[r][2 * q];
            const double fr = swap ? ar[r][2 * q] : ar[r][2 * q + 1], fi = swap ? ai[r][2 * q] : ai[r][2 * q + 1];
            double* o = ph + (kq + q) * ph_stride + d_phidx(A.lmax + 1, p, m);
            o[0] = kr + fr;
            o[1] = ki + fi;
            o[2] = kr - fr;
            o[3] = ki - fi;
        }
    }
}

// Adjoint lane state: G combinations of the Q and U phases
template <int R>
struct Adj2G {
    double qk_r[R], qk_i[R], qf_r[R], qf_i[R], uk_r[R], uk_i[R], uf_r[R], uf_i[R];
};

template <int R>
CMDR_HD void leg2_adj_load(const Leg2Args& A, const double* __restrict__ ph, int64_t ph_stride, int kq, int m,
                           int chunk, int lane, Adj2G<R>& G) {
    const int l0 = m > 2 ? m : 2;
    const bool swap = ((l0 + m) & 1) != 0;
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t idx = d_phidx(A.lmax + 1, base + r * 64, m);
        const double* q = ph + kq * ph_stride + idx;
        const double* u = ph + (kq + 1) * ph_stride + idx;
        // "keep" pairs with (N + S), "flip" with (N - S); for odd first-of-pair parity the roles swap
        const double qpr = q[0] + q[2], qpi = q[1] + q[3], qmr = q[0] - q[2], qmi = q[1] - q[3];
        const double upr = u[0] + u[2], upi = u[1] + u[3], umr = u[0] - u[2], umi = u[1] - u[3];
        G.qk_r[r] = swap ? qmr : qpr; G.qk_i[r] = swap ? qmi : qpi;
        G.qf_r[r] = swap ? qpr : qmr; G.qf_i[r] = swap ? qpi : qmi;
        G.uk_r[r] = swap ? umr : upr; G.uk_i[r] = swap ? umi : upi;
        G.uf_r[r] = swap ? upr : umr; G.uf_i[r] = swap ? upi : umi;
    }
}

// one group of 4 l (two pairs): v[4*j + {0..3}] = partial (E'r, E'i, B'r, B'i) of this lane for l = l0g + j
// transpose of leg2_synth_lane's accumulations:
//   first of pair : E' += W Gq_keep - i X Gu_flip... written out per component below
// a4[j] = alpha_{l0g + j + 1}, b4[j] = beta_{l0g + j + 1}
template <int R, bool INJECT>
CMDR_HD void leg2_adj_group_v(const double (&a4)[4], const double (&b4)[4], int l0g, Leg2State<R>& S, const Adj2G<R>& G,
                              double* v) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        const int l = l0g + j;
        double er = 0, ei = 0, br = 0, bi = 0, er2 = 0, ei2 = 0, br2 = 0, bi2 = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double W = S.pc[r] + S.mc[r], X = S.pc[r] - S.mc[r];
            // synthesis (first of pair): Qk += E' W ; Qf += i B' X ; Uk += B' W ; Uf += -i E' X
            er += W * G.qk_r[r] - X * G.uf_i[r];
            ei += W * G.qk_i[r] + X * G.uf_r[r];
            br += W * G.uk_r[r] + X * G.qf_i[r];
            bi += W * G.uk_i[r] - X * G.qf_r[r];
            leg2_advance<R, INJECT>(S, r, l, a4[j], b4[j]);
            W = S.pc[r] + S.mc[r];
            X = S.pc[r] - S.mc[r];
            // second of pair: Qf += E' W ; Qk += i B' X ; Uf += B' W ; Uk += -i E' X
            er2 += W * G.qf_r[r] - X * G.uk_i[r];
            ei2 += W * G.qf_i[r] + X * G.uk_r[r];
            br2 += W * G.uf_r[r] + X * G.qk_i[r];
            bi2 += W * G.uf_i[r] - X * G.qk_r[r];
            leg2_advance<R, INJECT>(S, r, l + 1, a4[j + 1], b4[j + 1]);
        }
        v[4 * j + 0] = er;  v[4 * j + 1] = ei;  v[4 * j + 2] = br;  v[4 * j + 3] = bi;
        v[4 * j + 4] = er2; v[4 * j + 5] = ei2; v[4 * j + 6] = br2; v[4 * j + 7] = bi2;
    }
}
template <int R, bool INJECT>
CMDR_HD void leg2_adj_group(const Leg2Args&, const double* __restrict__ al, const double* __restrict__ be, int l0g,
                            Leg2State<R>& S, const Adj2G<R>& G, double* v) {
    double a4[4], b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a4[j] = al[l0g + j + 1]; b4[j] = be[l0g + j + 1]; }
    leg2_adj_group_v<R, INJECT>(a4, b4, l0g, S, G, v);
}

// packed (E, B) a_lm -> spin-2 stream entry: E' = -E cnorm kappa_m / 2 (kappa = 1/sqrt2 for m > 0), l >= 2 only
CMDR_HD void alm2_to_stream_elem(const double* __restrict__ aE, const double* __restrict__ aB,
                                 double* __restrict__ st, int npol, int ip, const double* __restrict__ cnorm,
                                 int lmax, int m, int l) {
    const int64_t t = d_moffp(lmax, m) + (l - m);
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (l <= lmax && l >= 2) {
        const int64_t i = d_packed_index(lmax, l, m);
        const double f = -0.5 * cnorm[t] * (m == 0 ? 1.0 : 0.70710678118654752440);
        v[0] = aE[i] * f;
        v[2] = aB[i] * f;
        if (m > 0) { v[1] = aE[i + 1] * f; v[3] = aB[i + 1] * f; }
    }
    double* o = st + 4 * (t * npol + ip);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}

// spin-2 adjoint partial columns (4 doubles per l: E'r, E'i, B'r, B'i) -> packed E, B:
//   E = -cnorm kappa'_m / 2 * sum_chunks part ; kappa' = sqrt2 for m > 0 ; l < 2 -> 0
CMDR_HD void part2_to_alm_elem(const double* __restrict__ p, int64_t part_chunk_stride, int nchunk,
                               double* __restrict__ aE, double* __restrict__ aB, const double* __restrict__ cnorm,
                               int lmax, int m, int l, const int* __restrict__ lwtab = nullptr) {
    const int64_t t = d_moffp(lmax, m) + (l - m);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (l >= 2)
        for (int c = 0; c < nchunk; ++c) {
            if (lwtab && l < lwtab[m * nchunk + c]) continue;     // never written: structurally zero
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += p[c * part_chunk_stride + 4 * t + k];
        }
    const double f = -0.5 * cnorm[t] * (m == 0 ? 1.0 : 1.41421356237309504880);
    const int64_t i = d_packed_index(lmax, l, m);
    aE[i] = s[0] * f;
    aB[i] = s[2] * f;
    if (m > 0) { aE[i + 1] = s[1] * f; aB[i + 1] = s[3] * f; }
}

// ---------------------------------------------------------------------------------------------------------
// Ring FFT pieces.  buf = LDS image of M complex at padded positions lds_pad(i) (one 16-B pad per 8 elements, so
// that both "8 consecutive elements per lane" and "stride-h" register passes are bank-conflict free);
// tw = exp(+2 pi i k / Mmax), k < Mmax/2.  Transforms run as register passes of up to 3 radix-2 stages (radix-8
// butterflies): 12 stages = 4 passes = 4 barriers instead of 12.
struct FftCtx {
    int tid, nthr;
    // optional two-level twiddle table held in LDS (ring_tw_fill): exp(2 pi i k / Mmax) = hi[k >> s] * lo[k & (2^s - 1)],
    // so that the register passes carry no global load (one L2 round trip per pass and barrier otherwise)
    const cd* tw_hi = nullptr;
    const cd* tw_lo = nullptr;
    int tw_s = 0;
    int dbg = 0;     // timing experiments only (CMDR_RING_DEBUG_SKIP): 1 no phase loads, 2 no synthesis FFT, 4 no analysis FFT, 8 no stores
};
CMDR_HD cd fft_tw(const cd* __restrict__ tw, const FftCtx& c, int k) {
    if (!c.tw_hi) return tw[k];
    return cmul(c.tw_hi[k >> c.tw_s], c.tw_lo[k & ((1 << c.tw_s) - 1)]);
}
CMDR_HD int ring_tw_split(int log2Mmax) { return log2Mmax / 2; }                 // s: lo has 2^s entries, hi 2^(log2Mmax-1-s)
CMDR_HD int ring_tw_elems(int log2Mmax) { return (1 << ring_tw_split(log2Mmax)) + (1 << (log2Mmax - 1 - ring_tw_split(log2Mmax))); }
// fill the two-level table at dst (LDS); visible after the caller's next block barrier
CMDR_HD void ring_tw_fill(cd* dst, const cd* __restrict__ tw, int log2Mmax, FftCtx& c) {
    const int s = ring_tw_split(log2Mmax), nlo = 1 << s, nhi = 1 << (log2Mmax - 1 - s);
    for (int j = c.tid; j < nlo + nhi; j += c.nthr) dst[j] = j < nlo ? tw[j] : tw[(j - nlo) << s];
    c.tw_lo = dst;
    c.tw_hi = dst + nlo;
    c.tw_s = s;
}

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define CMDR_BLOCK_SYNC() __syncthreads()
#else
#define CMDR_BLOCK_SYNC() ((void)0)
#endif

CMDR_HD int lds_pad(int i) { return i + (i >> 3); }
CMDR_HD int lds_elems(int log2M) { return (1 << log2M) + (1 << log2M >> 3) + 1; }

// Twiddles of a register pass.  Stage t (half h << t) of the group at offset pos needs, for element j,
//   exp(2 pi i (pos + (j mod 2^t) h) / (2 h 2^t)) = b_t * exp(2 pi i (j mod 2^t) / 2^(t+1)),  b_t = exp(2 pi i pos / (h 2^(t+1)))
// so one table entry b_{K-1} per group gives every stage by squaring (b_{t-1} = b_t^2) and the second factor is one
// of 1, e^{i pi/4}, i, e^{3 i pi/4}: one global load per group instead of 2^K - 1.
CMDR_HD cd csqr(cd a) { return {a.x * a.x - a.y * a.y, 2.0 * a.x * a.y}; }
template <int T>
CMDR_HD cd rot_const(cd v, int jl) {   // v * exp(2 pi i jl / 2^(T+1)), jl < 2^T, T <= 2
    if (T == 0) return v;
    if (T == 1) return jl ? cd{-v.y, v.x} : v;
    constexpr double r = 0.70710678118654752440;
    switch (jl) {
        case 0: return v;
        case 1: return {(v.x - v.y) * r, (v.x + v.y) * r};
        case 2: return {-v.y, v.x};
        default: return {-(v.x + v.y) * r, (v.x - v.y) * r};
    }
}

// K radix-2 DIT stages (halves h, 2h, .. h<<(K-1), h = 1<<hl) on the 2^K elements i0 + j*h held in registers.
// bit-reversed order in -> natural order out, kernel exp(+2 pi i jk/M)
// post (last pass only): a pointwise step rides on the store that leaves the transform in natural order --
// buf[i] = post(i, value) -- instead of costing its own LDS round trip and barrier
struct NoPost {
    CMDR_HD cd load(int) const { return {0.0, 0.0}; }
    CMDR_HD cd apply(cd, cd v) const { return v; }
};
template <int K, class Post = NoPost>
CMDR_HD void fft_dit_pass(cd* buf, int log2M, int hl, const cd* __restrict__ tw, int log2Mmax, FftCtx c,
                          const Post& post = Post{}, bool last = false) {
    constexpr int N = 1 << K;
    const int ngroups = 1 << (log2M - K);
    const int hmask = (1 << hl) - 1;
    for (int g = c.tid; g < ngroups; g += c.nthr) {
        const int pos = g & hmask;
        const int i0 = ((g >> hl) << (hl + K)) + pos;
        cd bt[K];
        bt[K - 1] = fft_tw(tw, c, pos << (log2Mmax - (hl + K)));
#pragma unroll
        for (int t = K - 1; t > 0; --t) bt[t - 1] = csqr(bt[t]);
        cd v[N], pre[N];
        if (last) {   // the pointwise operands (global memory) are requested before the butterflies, used at the store
#pragma unroll
            for (int j = 0; j < N; ++j) pre[j] = post.load(i0 + (j << hl));
        }
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = buf[lds_pad(i0 + (j << hl))];
#pragma unroll
        for (int t = 0; t < K; ++t) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j & (1 << t)) continue;
                const cd u = v[j];
                cd x = cmul(v[j + (1 << t)], bt[t]);
                x = t == 0 ? x : (t == 1 ? rot_const<1>(x, j & 1) : rot_const<2>(x, j & 3));
                v[j] = cadd(u, x);
                v[j + (1 << t)] = csub(u, x);
            }
        }
        if (last) {
#pragma unroll
            for (int j = 0; j < N; ++j) buf[lds_pad(i0 + (j << hl))] = post.apply(pre[j], v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) buf[lds_pad(i0 + (j << hl))] = v[j];
        }
    }
    CMDR_BLOCK_SYNC();
}

// K radix-2 DIF stages (halves h<<(K-1), .., 2h, h).  natural order in -> bit-reversed order out.
// post (optional, last pass only): store conj(v) * post[position] instead of v -- the Bluestein spectrum product
template <int K>
CMDR_HD void fft_dif_pass(cd* buf, int log2M, int hl, const cd* __restrict__ tw, int log2Mmax, FftCtx c,
                          const cd* __restrict__ post = nullptr) {
    constexpr int N = 1 << K;
    const int ngroups = 1 << (log2M - K);
    const int hmask = (1 << hl) - 1;
    for (int g = c.tid; g < ngroups; g += c.nthr) {
        const int pos = g & hmask;
        const int i0 = ((g >> hl) << (hl + K)) + pos;
        cd bt[K];
        bt[K - 1] = fft_tw(tw, c, pos << (log2Mmax - (hl + K)));
#pragma unroll
        for (int t = K - 1; t > 0; --t) bt[t - 1] = csqr(bt[t]);
        cd v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = buf[lds_pad(i0 + (j << hl))];
#pragma unroll
        for (int t = K - 1; t >= 0; --t) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j & (1 << t)) continue;
                const cd u = v[j], x = v[j + (1 << t)];
                v[j] = cadd(u, x);
                cd d = cmul(csub(u, x), bt[t]);
                v[j + (1 << t)] = t == 0 ? d : (t == 1 ? rot_const<1>(d, j & 1) : rot_const<2>(d, j & 3));
            }
        }
        if (post) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int i = i0 + (j << hl);
                buf[lds_pad(i)] = cmul(cconj(v[j]), post[i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) buf[lds_pad(i0 + (j << hl))] = v[j];
        }
    }
    CMDR_BLOCK_SYNC();
}

template <class Post = NoPost>
CMDR_HD void fft_dit_plus(cd* buf, int log2M, const cd* __restrict__ tw, int log2Mmax, FftCtx c,
                          const Post& post = Post{}) {
    int hl = 0;
    const int k0 = log2M % 3;
    if (k0 == 1) { fft_dit_pass<1, Post>(buf, log2M, 0, tw, log2Mmax, c, post, log2M == 1); hl = 1; }
    else if (k0 == 2) { fft_dit_pass<2, Post>(buf, log2M, 0, tw, log2Mmax, c, post, log2M == 2); hl = 2; }
    for (; hl < log2M; hl += 3) fft_dit_pass<3, Post>(buf, log2M, hl, tw, log2Mmax, c, post, hl + 3 == log2M);
}

CMDR_HD void fft_dif_plus(cd* buf, int log2M, const cd* __restrict__ tw, int log2Mmax, FftCtx c,
                          const cd* __restrict__ post = nullptr) {
    const int k0 = log2M % 3;
    for (int hl = log2M - 3; hl >= k0; hl -= 3)
        fft_dif_pass<3>(buf, log2M, hl, tw, log2Mmax, c, (hl == 0) ? post : nullptr);
    if (k0 == 1) fft_dif_pass<1>(buf, log2M, 0, tw, log2Mmax, c, post);
    else if (k0 == 2) fft_dif_pass<2>(buf, log2M, 0, tw, log2Mmax, c, post);
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
    long long chirp_off;   // per-length table: rot[n] (, w[nt], chat[M] for Bluestein; nt = n or n/2 if split)
    int ring;
    int split;             // 0, or 1 + index of this pair's scratch line (ring transformed as two half-length pieces)
    int log2T;             // 0, or log2 of the circulant size of this pair's Toeplitz form of Y^t diag(mul) Y (below)
    long long that_off;    // offset of the pair's multiplier spectrum in a per-map array (complex units)
};

// Masked monopole / dipole sums over the pixels of one ring (applyMonoDipolePrior, comm_diffuse_comp_mod.f90:5761-5794).
//   type 1 ('monopole'):         acc[0] += map * mask, acc[1] += mask                       (:5764-5765; mask as weight)
//   type 2 ('monopole+dipole'):  pixels with mask >= 0.5 (:5781): v = (1, pix2vec_ring), acc[0..9] += upper triangle of
//                                v v^t (row-major: 00 01 02 03 11 12 13 22 23 33), acc[10..13] += v * map   (:5782-5790)
// z of the ring from its number as HEALPix pix2vec_ring forms it (cap: 1 - i^2 / (3 N^2), belt: (2 N - i) 2 / (3 N));
// phi of pixel k = phi0 + 2 pi k / nphi.  `south` mirrors the ring (z -> -z).
constexpr int kMdSums = 16;
CMDR_HD double healpix_ring_z(int nside, int ring /* northern ring number 1..2 nside */) {
    if (ring < nside) return 1.0 - (double)ring * ring / (3.0 * nside * (double)nside);
    return (2.0 * nside - ring) * 2.0 / (3.0 * nside);
}
CMDR_HD void md_pixel_accum(int type, double z, double sth, double phi, double mapv, double maskv, double* acc) {
    if (type == 1) {
        acc[0] += mapv * maskv;
        acc[1] += maskv;
        return;
    }
    if (maskv < 0.5) return;
    const double v1 = sth * cos(phi), v2 = sth * sin(phi), v3 = z;
    acc[0] += 1.0;     acc[1] += v1;      acc[2] += v2;      acc[3] += v3;
    acc[4] += v1 * v1; acc[5] += v1 * v2; acc[6] += v1 * v3;
    acc[7] += v2 * v2; acc[8] += v2 * v3; acc[9] += v3 * v3;
    acc[10] += mapv;   acc[11] += v1 * mapv; acc[12] += v2 * mapv; acc[13] += v3 * mapv;
}

// HEALPix rings have n*phi0 = pi (phi0 = pi/(4i), n = 4i; belt: pi/(4N), n = 4N) or phi0 = 0, hence
//   e^{i (j + k n) phi0} = (-1)^k rot_j ,  e^{i (k n - j) phi0} = (-1)^k conj(rot_j) ,  rot_j = e^{i pi j / n}
// and one tabulated rotation per slot replaces every sincos.  For phi0 = 0 the (-1)^k signs disappear.
//
// Packed spectrum slot j of a ring pair from its phases (gather form, no atomics):
//   Z_j = X^N_j + i X^S_j ,  X_j = rot_j [ sum_k s^k F_{j+kn} + sum_{k>=1} s^k conj F_{kn-j} ],  s = -1 (or +1)
CMDR_HD cd ring_gather_slot(const double* __restrict__ ph, int64_t prow, int pair, int n, int mmax, cd rot,
                            bool flip, int j) {
    double nr = 0.0, ni = 0.0, sr = 0.0, si = 0.0;
    double sg = 1.0;
    for (int m = j; m <= mmax; m += n) {  // direct aliases
        const double* f = ph + d_phidx(prow, pair, m);
        nr += sg * f[0]; ni += sg * f[1]; sr += sg * f[2]; si += sg * f[3];
        if (flip) sg = -sg;
    }
    sg = flip ? -1.0 : 1.0;
    for (int m = (j == 0 ? n : n - j); m <= mmax; m += n) {  // conjugate aliases (m > 0)
        const double* f = ph + d_phidx(prow, pair, m);
        nr += sg * f[0]; ni -= sg * f[1]; sr += sg * f[2]; si -= sg * f[3];
        if (flip) sg = -sg;
    }
    const cd gn = cmul({nr, ni}, rot), gs = cmul({sr, si}, rot);
    return {gn.x - gs.y, gn.y + gs.x};   // gn + i gs
}

// Scatter form of the spectrum build for rings without aliasing (n > 2 mmax: the belt and the larger cap rings, i.e.
// almost all of the work): every m owns slot m and slot n-m exclusively, so each thread streams its m values with
// independent (unrolled) global loads instead of a dependent gather per slot.
//   BLUE: slot value is conj(Z_j w_j) at natural position (Bluestein input); else Z_j at the bit-reversed position.
template <bool BLUE>
CMDR_HD void ring_scatter(cd* buf, const RingDev& d, const double* __restrict__ ph, int64_t prow, int pair,
                          const cd* __restrict__ rot, const cd* __restrict__ w, FftCtx c) {
    const int n = d.nphi, M = 1 << d.log2M, mmax = d.mmax_eff;
    const bool flip = d.phi0 != 0.0;
    // slots nobody writes below: mmax < j < n - mmax, and the Bluestein padding n <= j < M  (disjoint: no barrier)
    for (int j = mmax + 1 + c.tid; j < n - mmax; j += c.nthr) buf[lds_pad(BLUE ? j : d_bitrev(j, d.log2M))] = {0.0, 0.0};
    if (BLUE)
        for (int j = n + c.tid; j < M; j += c.nthr) buf[lds_pad(j)] = {0.0, 0.0};
    constexpr int U = 4;
    for (int m0 = c.tid; m0 <= mmax; m0 += U * c.nthr) {
        double f[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + u * c.nthr;
            const double* q = ph + d_phidx(prow, pair, m <= mmax ? m : mmax);
            f[u][0] = q[0]; f[u][1] = q[1]; f[u][2] = q[2]; f[u][3] = q[3];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + u * c.nthr;
            if (m > mmax) continue;
            const cd r = flip ? rot[m] : cd{1.0, 0.0};
            const cd gn = cmul({f[u][0], f[u][1]}, r), gs = cmul({f[u][2], f[u][3]}, r);
            cd z = {gn.x - gs.y, gn.y + gs.x};          // X^N_m + i X^S_m
            cd zc = {gn.x + gs.y, gs.x - gn.y};         // conj(gn) + i conj(gs)  -> slot n - m
            if (BLUE) {
                buf[lds_pad(m)] = cconj(cmul(z, w[m]));
                if (m > 0) buf[lds_pad(n - m)] = cconj(cmul(zc, w[n - m]));
            } else {
                buf[lds_pad(d_bitrev(m, d.log2M))] = z;
                if (m > 0) buf[lds_pad(d_bitrev(n - m, d.log2M))] = zc;
            }
        }
    }
    CMDR_BLOCK_SYNC();
}

// One length-n DFT inside the LDS image: n a power of two (log2M = log2 n) or Bluestein (M >= 2n-1) with the chirp
// w_j = e^{i pi j^2/n} (j < n) and chat = bit-reversed FFT_M^- of the conjugate chirp.
struct FftSub {
    int n, log2M, bluestein;
    const cd* w;
    const cd* chat;
};

// Where / in which form the inverse transform wants spectrum slot j:
//   power of two: value at the bit-reversed position; Bluestein: conj(value * w_j) at the natural position
//   (positions n..M-1 must hold zeros).
CMDR_HD void idft_put(cd* buf, const FftSub& f, int j, cd v) {
    if (f.bluestein) buf[lds_pad(j)] = cconj(cmul(v, f.w[j]));
    else buf[lds_pad(d_bitrev(j, f.log2M))] = v;
}

// Inverse DFT y_k = sum_j Z_j e^{+2 pi i jk/n} of slots placed with idft_put().  The result is read with idft_at():
// the closing Bluestein factor w_k / M is applied on read, so that callers fold it into whatever pass touches the
// pixels next instead of paying an LDS round trip for it.
CMDR_HD void idft_core(cd* buf, const FftSub& f, const cd* __restrict__ tw, int log2Mmax, FftCtx c) {
    if (!f.bluestein) {
        fft_dit_plus(buf, f.log2M, tw, log2Mmax, c);
        return;
    }
    // a_j = Z_j w_j ; F^-(a) = conj(F^+(conj a)); the product with the chirp spectrum rides on the last DIF store
    fft_dif_plus(buf, f.log2M, tw, log2Mmax, c, f.chat);
    fft_dit_plus(buf, f.log2M, tw, log2Mmax, c);
}
CMDR_HD cd idft_at(const cd* buf, const FftSub& f, int k) {
    const cd v = buf[lds_pad(k)];
    if (!f.bluestein) return v;
    const double inv = 1.0 / (double)(1 << f.log2M);
    const cd r = cmul(v, f.w[k]);
    return {r.x * inv, r.y * inv};
}

// Forward DFT Z_j = sum_k z_k e^{-2 pi i jk/n}.  Input form (dft_put, k < n; Bluestein images need zeros in n..M-1):
// power of two conj(z_k), Bluestein z_k conj(w_k).  Read the result with dft_at().
CMDR_HD cd dft_in(const FftSub& f, int k, cd z) {
    if (!f.bluestein) return cconj(z);
    return cmul(z, cconj(f.w[k]));        // conj(a_k), a_k = conj(z_k) w_k
}
CMDR_HD void dft_core(cd* buf, const FftSub& f, const cd* __restrict__ tw, int log2Mmax, FftCtx c) {
    if (!f.bluestein) {
        fft_dif_plus(buf, f.log2M, tw, log2Mmax, c);   // Z = conj(F^+(conj z)); DIF leaves it bit-reversed
    } else {
        // conj(Z)_j = w_j sum_k (conj(z_k) w_k) conj(w_{j-k})  (same chirp machinery)
        fft_dif_plus(buf, f.log2M, tw, log2Mmax, c, f.chat);
        fft_dit_plus(buf, f.log2M, tw, log2Mmax, c);
    }
}
CMDR_HD cd dft_at(const cd* buf, const FftSub& f, int j) {
    if (!f.bluestein) return cconj(buf[lds_pad(d_bitrev(j, f.log2M))]);
    const double inv = 1.0 / (double)(1 << f.log2M);
    const cd v = cmul(buf[lds_pad(j)], f.w[j]);          // conj(Z_j) * M
    return {v.x * inv, -v.y * inv};
}

CMDR_HD FftSub ring_fft_desc(const RingDev& d, const cd* __restrict__ chirp) {
    // per-length table: rot[n], then for the transform length nt (= n, or n/2 for split rings): w[nt], chat[M]
    const cd* w = chirp + d.chirp_off + d.nphi;
    const int nt = d.split ? d.nphi / 2 : d.nphi;
    return FftSub{nt, d.log2M, d.bluestein, w, w + nt};
}

// Full inverse (synthesis) ring transform in LDS: on return buf[lds_pad(k)], k<n holds y^N_k + i y^S_k.
struct PixelPost {           // (y_N, y_S) -> conj(mul_N y_N, mul_S y_S); no southern ring: imaginary part 0
    const double* __restrict__ a;
    const double* __restrict__ b;
    CMDR_HD cd load(int k) const { return {a[k], b ? b[k] : 0.0}; }
    CMDR_HD cd apply(cd m, cd z) const { return {z.x * m.x, -(z.y * m.y)}; }
};
CMDR_HD void ring_synth_lds(cd* buf, const RingDev& d, const double* __restrict__ ph, int64_t prow, int pair,
                            const cd* __restrict__ tw, int log2Mmax, const cd* __restrict__ chirp, FftCtx c,
                            bool fuse_mul = false, const double* __restrict__ mulN = nullptr,
                            const double* __restrict__ mulS = nullptr) {
    const int n = d.nphi, M = 1 << d.log2M;
    const bool flip = d.phi0 != 0.0;
    const cd* rot = chirp + d.chirp_off;   // rot_j, j < n
    const FftSub f = ring_fft_desc(d, chirp);
    if (n > 2 * d.mmax_eff) {
        if (d.bluestein) ring_scatter<true>(buf, d, ph, prow, pair, rot, f.w, c);
        else ring_scatter<false>(buf, d, ph, prow, pair, rot, nullptr, c);
    } else {
        for (int j = c.tid; j < (d.bluestein ? M : n); j += c.nthr) {
            if (j < n) {
                const cd r = flip ? rot[j] : cd{1.0, 0.0};
                idft_put(buf, f, j, ring_gather_slot(ph, prow, pair, n, d.mmax_eff, r, flip, j));
            } else {
                buf[lds_pad(j)] = {0.0, 0.0};
            }
        }
        CMDR_BLOCK_SYNC();
    }
    if (fuse_mul) fft_dit_plus(buf, f.log2M, tw, log2Mmax, c, PixelPost{mulN, mulS});   // power-of-two rings only
    else idft_core(buf, f, tw, log2Mmax, c);
}

// Extract G^N_m, G^S_m (m <= mmax_eff) from the packed spectrum Z_j = spec(j) and store them as phases for the
// adjoint Legendre stage: X^N_j = (Z_j + conj Z_{n-j})/2, X^S_j = (Z_j - conj Z_{n-j})/(2i);
// G_{j+kn} = X_j e^{-i m phi0} = X_j s^k conj(rot_j).
template <class Spec>
CMDR_HD void ring_store_phases(const Spec& spec, const RingDev& d, double* __restrict__ ph, int64_t prow,
                               int pair, const cd* __restrict__ chirp, FftCtx c) {
    const int n = d.nphi;
    const bool flip = d.phi0 != 0.0;
    const cd* rot = chirp + d.chirp_off;
    const int jmax = d.mmax_eff < n - 1 ? d.mmax_eff : n - 1;
    for (int j = c.tid; j <= jmax; j += c.nthr) {
        const cd a = spec(j), b = cconj(spec(j == 0 ? 0 : n - j));
        const cd xn = {0.5 * (a.x + b.x), 0.5 * (a.y + b.y)};
        const cd dm = {0.5 * (a.x - b.x), 0.5 * (a.y - b.y)};
        const cd xs = {dm.y, -dm.x};  // dm / i
        const cd e = flip ? cconj(rot[j]) : cd{1.0, 0.0};
        cd gn = cmul(xn, e), gs = cmul(xs, e);
        for (int m = j; m <= d.mmax_eff; m += n) {
            double* o = ph + d_phidx(prow, pair, m);
            o[0] = gn.x; o[1] = gn.y; o[2] = gs.x; o[3] = gs.y;
            if (flip) { gn.x = -gn.x; gn.y = -gn.y; gs.x = -gs.x; gs.y = -gs.y; }
        }
    }
}

struct SpecDirect {   // spectrum of a ring transformed in one piece
    const cd* buf;
    FftSub f;
    CMDR_HD cd operator()(int j) const { return dft_at(buf, f, j); }
};

// e^{+2 pi i j / n} from rot_t = e^{i pi t / n}, t < n
CMDR_HD cd ring_unit(const cd* __restrict__ rot, int n, int j) {
    const int t = 2 * j;
    if (t < n) return rot[t];
    const cd r = rot[t - n];
    return {-r.x, -r.y};
}

// Split rings (n even, too long for one LDS image): radix-2 decimation in pixel space,
//   y_{k1+2k2} = sum_{j2<h} e^{2 pi i j2 k2/h} V_{k1}[j2],  V_{k1}[j2] = e^{2 pi i j2 k1/n} (Z_{j2} + (-1)^{k1} Z_{h+j2})
//   Z_j = A_0[j mod h] + e^{-2 pi i j/n} A_1[j mod h],       A_{k1}[j'] = sum_{k2} z_{k1+2k2} e^{-2 pi i j' k2/h}
// with h = n/2: two half-length transforms run one after the other in the same LDS image; A_0 waits in a
// per-workgroup global scratch line (h complex) until A_1 exists.
struct SpecSplit {
    const cd* buf;
    FftSub f;
    const cd* a0;
    const cd* rot;
    int n;
    CMDR_HD cd operator()(int j) const {
        const int h = n >> 1, jj = j < h ? j : j - h;
        const cd e = cconj(ring_unit(rot, n, j));
        const cd t = cmul(e, dft_at(buf, f, jj));
        const cd a = a0[jj];
        return {a.x + t.x, a.y + t.y};
    }
};

CMDR_HD cd ring_split_input(const double* __restrict__ ph, int64_t prow, int pair, const RingDev& d,
                            const cd* __restrict__ rot, bool flip, int k1, int j2) {
    const int n = d.nphi, h = n >> 1;
    const cd one = {1.0, 0.0};
    const cd a = ring_gather_slot(ph, prow, pair, n, d.mmax_eff, flip ? rot[j2] : one, flip, j2);
    const cd b = ring_gather_slot(ph, prow, pair, n, d.mmax_eff, flip ? rot[h + j2] : one, flip, h + j2);
    if (k1 == 0) return cadd(a, b);
    return cmul(csub(a, b), rot[2 * j2]);
}

// ---------------------------------------------------------------------------------------------------------
// Toeplitz form of the fused ring operator (mode 2) for rings whose length is not a power of two.
// On one ring  G_m = sum_k mul_k e^{-i m phi_k} sum_{m'} F_m' e^{+i m' phi_k} = sum_{m'} t_{m-m'} F_m' ,
//   t_d = sum_k mul_k e^{-i d phi_k}   (|d| <= 2 mmax; t_{-d} = conj t_d; F_{-m} = conj F_m),
// a Toeplitz product, i.e. a circular convolution of any length M >= 4 mmax + 1 -- a power of two, so the two
// Bluestein transforms per direction (4 FFTs of M_B >= 2 n - 1) become 2 radix-2 FFTs of M_T <= M_B, with no chirp
// products and no e^{i m phi0} rotations (they are inside t_d).  The pixel multiplier is replaced by its circulant
// spectrum  tau(k) / M = (1/M) sum_d t_d e^{2 pi i d k / M}  (real), built once per multiplier map by
// ring_toeplitz_spec from t_d (= the mode-1 transform of the multiplier map itself, run up to 2 mmax).
// North and south ring share one complex FFT exactly as in the pixel form (real part north, imaginary part south).
CMDR_HD void ring_toeplitz_load(cd* buf, int lg, int mmax, const double* __restrict__ ph, int64_t prow, int pair,
                                FftCtx c) {
    const int M = 1 << lg;
    for (int j = mmax + 1 + c.tid; j < M - mmax; j += c.nthr) buf[lds_pad(d_bitrev(j, lg))] = {0.0, 0.0};
    constexpr int U = 4;
    for (int m0 = c.tid; m0 <= mmax; m0 += U * c.nthr) {
        double f[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + u * c.nthr;
            const double* q = ph + d_phidx(prow, pair, m <= mmax ? m : mmax);
            f[u][0] = q[0]; f[u][1] = q[1]; f[u][2] = q[2]; f[u][3] = q[3];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + u * c.nthr;
            if (m > mmax) continue;
            buf[lds_pad(d_bitrev(m, lg))] = {f[u][0] - f[u][3], f[u][1] + f[u][2]};                    // F^N_m + i F^S_m
            if (m > 0) buf[lds_pad(d_bitrev(M - m, lg))] = {f[u][0] + f[u][3], f[u][2] - f[u][1]};     // conj F^N + i conj F^S
        }
    }
    CMDR_BLOCK_SYNC();
}

struct ToeplitzPost {
    const cd* __restrict__ T;
    CMDR_HD cd load(int k) const { return T[k]; }
    CMDR_HD cd apply(cd t, cd z) const { return {z.x * t.x, -(z.y * t.y)}; }
};
// phases (in place) -> G = T F for one ring pair; that = this map's multiplier spectra
CMDR_HD void ring_toeplitz_apply(cd* buf, const RingDev& d, double* __restrict__ ph, int64_t prow, int pair,
                                 const cd* __restrict__ that, const cd* __restrict__ tw, int log2Mmax, FftCtx c) {
    const int lg = d.log2T, M = 1 << lg, mmax = d.mmax_eff;
    (void)M;
    ring_toeplitz_load(buf, lg, mmax, ph, prow, pair, c);
    // x(k) = sum_j Z_j e^{+2 pi i jk/M}: (north, south) real signals; the multiplier spectrum rides on the last pass's
    // store, which leaves conj(tau(k) x(k)): the input form of the forward transform
    fft_dit_plus(buf, lg, tw, log2Mmax, c, ToeplitzPost{that + d.that_off});
    fft_dif_plus(buf, lg, tw, log2Mmax, c);                       // Y_j = conj(buf[bitrev j])
    for (int j = c.tid; j <= mmax; j += c.nthr) {
        const cd a = cconj(buf[lds_pad(d_bitrev(j, lg))]);
        const cd b = buf[lds_pad(d_bitrev(j == 0 ? 0 : M - j, lg))];   // conj(Y_{M-j})
        double* o = ph + d_phidx(prow, pair, j);
        o[0] = 0.5 * (a.x + b.x); o[1] = 0.5 * (a.y + b.y);           // G^N_j = (Y_j + conj Y_{M-j}) / 2
        o[2] = 0.5 * (a.y - b.y); o[3] = -0.5 * (a.x - b.x);          // G^S_j = (Y_j - conj Y_{M-j}) / (2i)
    }
}

// multiplier spectrum of one ring pair: td = phase-layout array holding t_d (north, south) for d <= 2 mmax
CMDR_HD void ring_toeplitz_spec(cd* buf, const RingDev& d, const double* __restrict__ td, int64_t prow, int pair,
                                cd* __restrict__ that, const cd* __restrict__ tw, int log2Mmax, FftCtx c) {
    const int lg = d.log2T, M = 1 << lg;
    ring_toeplitz_load(buf, lg, 2 * d.mmax_eff, td, prow, pair, c);
    fft_dit_plus(buf, lg, tw, log2Mmax, c);                       // tau^N(k) + i tau^S(k), both real
    const double inv = 1.0 / (double)M;
    cd* __restrict__ T = that + d.that_off;
    for (int k = c.tid; k < M; k += c.nthr) {
        const cd v = buf[lds_pad(k)];
        T[k] = {v.x * inv, v.y * inv};
    }
}

// Whole ring-pair job of one workgroup.  MODE 0: phases -> map (* mul * weight); 1: map (* mul * weight) -> phases;
// 2: phases -> pixels * mul -> phases (the fused Y^t N^-1 Y core of the matvec; the map never exists in HBM).
// Every pointwise step (Bluestein's closing chirp factor, the pixel multiplier, the analysis input form, zero padding)
// is folded into the one pass that touches the pixels.
template <int MODE>
CMDR_HD void ring_pixels(cd* buf, const FftSub& f, int npix, int k1, int kstep, const RingDev& d,
                         double* __restrict__ mp, const double* __restrict__ mu, double wg, FftCtx c) {
    // LDS element k2 <-> ring pixel k = k1 + kstep * k2 (kstep = 2 for the halves of a split ring)
    const int M = 1 << f.log2M;
    if (MODE == 0) {
        for (int k2 = c.tid; k2 < npix; k2 += c.nthr) {
            const int k = k1 + kstep * k2;
            const cd v = idft_at(buf, f, k2);
            mp[d.startN + k] = v.x * (wg * (mu ? mu[d.startN + k] : 1.0));
            if (d.startS >= 0) mp[d.startS + k] = v.y * (wg * (mu ? mu[d.startS + k] : 1.0));
        }
        return;
    }
    for (int k2 = c.tid; k2 < (f.bluestein ? M : npix); k2 += c.nthr) {
        cd o = {0.0, 0.0};
        if (k2 < npix) {
            const int k = k1 + kstep * k2;
            cd z;
            if (MODE == 1) {
                z = {mp[d.startN + k] * (wg * (mu ? mu[d.startN + k] : 1.0)), 0.0};
                if (d.startS >= 0) z.y = mp[d.startS + k] * (wg * (mu ? mu[d.startS + k] : 1.0));
            } else {
                z = idft_at(buf, f, k2);
                z.x *= mu[d.startN + k];
                z.y = d.startS >= 0 ? z.y * mu[d.startS + k] : 0.0;
            }
            o = dft_in(f, k2, z);
        }
        buf[lds_pad(k2)] = o;
    }
    CMDR_BLOCK_SYNC();
}

template <int MODE>
CMDR_HD void ring_block(cd* buf, const RingDev& d, int pair, double* __restrict__ php, int64_t prow,
                        double* __restrict__ mp, const double* __restrict__ mu, double wg,
                        const cd* __restrict__ tw, int log2Mmax, const cd* __restrict__ chirp,
                        cd* __restrict__ scratch, FftCtx c, const cd* __restrict__ that = nullptr) {
    if (MODE == 2 && that && d.log2T) {
        ring_toeplitz_apply(buf, d, php, prow, pair, that, tw, log2Mmax, c);
        return;
    }
    const int n = d.nphi;
    const FftSub f = ring_fft_desc(d, chirp);
    if (!d.split) {
        if (MODE == 2 && !d.bluestein) {
            // power-of-two ring of the fused pass: the pixel multiplier (and the conjugation the forward transform
            // wants) ride on the last synthesis pass's store; no separate pixel pass
            if (!(c.dbg & 3))
                ring_synth_lds(buf, d, php, prow, pair, tw, log2Mmax, chirp, c, true, mu + d.startN,
                               d.startS >= 0 ? mu + d.startS : nullptr);
            else if (!(c.dbg & 2)) fft_dit_plus(buf, f.log2M, tw, log2Mmax, c, PixelPost{mu + d.startN, d.startS >= 0 ? mu + d.startS : nullptr});
            if (!(c.dbg & 4)) dft_core(buf, f, tw, log2Mmax, c);
            if (!(c.dbg & 8)) ring_store_phases(SpecDirect{buf, f}, d, php, prow, pair, chirp, c);
            return;
        }
        if (MODE == 0 || MODE == 2) ring_synth_lds(buf, d, php, prow, pair, tw, log2Mmax, chirp, c);
        ring_pixels<MODE>(buf, f, n, 0, 1, d, mp, mu, wg, c);
        if (MODE == 0) return;
        dft_core(buf, f, tw, log2Mmax, c);
        ring_store_phases(SpecDirect{buf, f}, d, php, prow, pair, chirp, c);
        return;
    }
    const int h = n >> 1, M = 1 << d.log2M;
    const bool flip = d.phi0 != 0.0;
    const cd* rot = chirp + d.chirp_off;
    for (int k1 = 0; k1 < 2; ++k1) {
        if (MODE == 0 || MODE == 2) {
            for (int j = c.tid; j < (f.bluestein ? M : h); j += c.nthr) {
                if (j < h) idft_put(buf, f, j, ring_split_input(php, prow, pair, d, rot, flip, k1, j));
                else buf[lds_pad(j)] = {0.0, 0.0};
            }
            CMDR_BLOCK_SYNC();
            idft_core(buf, f, tw, log2Mmax, c);
        }
        ring_pixels<MODE>(buf, f, h, k1, 2, d, mp, mu, wg, c);
        if (MODE == 0) {
            CMDR_BLOCK_SYNC();
            continue;
        }
        dft_core(buf, f, tw, log2Mmax, c);
        if (k1 == 0) {
            for (int j = c.tid; j < h; j += c.nthr) scratch[j] = dft_at(buf, f, j);
            CMDR_BLOCK_SYNC();
        }
    }
    ring_store_phases(SpecSplit{buf, f, scratch, rot, n}, d, php, prow, pair, chirp, c);
}

// ---------------------------------------------------------------------------------------------------------
// a_lm layout conversion elements (one (l, m) each).
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
                              double* __restrict__ a, const double* __restrict__ cnorm, int lmax, int m, int l,
                              const int* __restrict__ lwtab = nullptr) {
    const int64_t t = d_moffp(lmax, m) + (l - m);
    double re = 0.0, im = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        if (lwtab && l < lwtab[m * nchunk + c]) continue;     // never written: structurally zero
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
