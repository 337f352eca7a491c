// Host-side (HIP-free) tables of an SHT plan: Legendre recursion coefficients, per-(m, ring-pair) start seeds,
// wave task list, ring/FFT descriptors and Bluestein chirps.  Built once per (nside, lmax, ring subset).
//
// Replaces the libsharp2 info objects Commander creates in commander3/src/comm_map_mod.f90:264-283
// (sharp_make_mmajor_real_packed_alm_info / sharp_make_healpix_geom_info).
#pragma once
#include <cstdint>
#include <vector>

#include "geom.hpp"

namespace cmdr {

constexpr int kWave = 64;
constexpr int kLsNever = 0x3fffffff;  // "this (m, pair) never starts"
constexpr int kAdjL = 8;              // l per transpose-reduce group in the adjoint Legendre kernel

// Padded triangle: column m holds l = m..lmax+1 (one zero pad entry), so kernels may run l in (even, odd) pairs.
inline int64_t moffp(int lmax, int m) { return (int64_t)m * (lmax + 2) - (int64_t)m * (m - 1) / 2; }
inline int64_t ntrip(int lmax) { return moffp(lmax, lmax + 1) + 64; }  // + slack for look-ahead reads (k_leg_adj_mx: up to l+32)

struct WaveTask {  // one wavefront's work item: 64*R colatitude pairs of one m
    int m, chunk, lw, lAend;
};

// Legendre stage over north/south-symmetric colatitude pairs (HEALPix ring pairs, or Gauss-Legendre nodes).
struct LegendreTables {
    int lmax = -1, npair = 0, R = 1, npair_pad = 0, nchunk = 0;
    std::vector<double> x, sth;         // [npair_pad] cos/sin(theta) of the northern member (x >= 0)
    std::vector<int> mlim;              // [npair_pad] largest m with non-negligible lambda (-1 for padding)
    std::vector<double> alpha;          // [ntrip] mu_l = alpha_l x mu_{l-1} - mu_{l-2}
    std::vector<double> cnorm;          // [ntrip] lambda_lm = cnorm * mu_l  (pad entries 0)
    std::vector<int> ls;                // [(lmax+1) * npair_pad] first l with |mu| above threshold
    std::vector<double> seedc, seedp;   // mu_{ls}, mu_{ls-1}
    bool uniform_start = false;         // every 64-pair lane block switches on at one l == m (mod 32) (plan_tables.cpp)
    // adjoint kernel: R pairs per lane, one task per wavefront, 4 per workgroup, longest first
    std::vector<WaveTask> tasks;
    std::vector<int> lw_chunk;           // [(lmax+1) * nchunk] first l the adjoint writes for (m, chunk); lmax+2: nothing
    // the same adjoint tasks as two lists, m < m_split | m >= m_split (each longest first, padded to a multiple of 4),
    // of about equal work: the first half's output can be summed over the ranks while the second half computes
    std::vector<WaveTask> tasks_split;
    int m_split = 0, nsplit_lo = 0;      // tasks_split[0 .. nsplit_lo) = the m < m_split part
    int group = 4;
    bool synth_wg = false;               // tasks_s grouped: tasks_s[4i .. 4i+3] = 4 chunks of one m (chunk = -1: none)
    // synthesis kernel: Rs pairs per lane, same layout (its own list so R and Rs can be tuned independently)
    int Rs = 1, nchunk_s = 0;
    std::vector<WaveTask> tasks_s;
    void build(int lmax, const std::vector<double>& x, const std::vector<double>& sth, int R, int Rs,
               int nthreads = 0, const std::vector<int>* mlim_in = nullptr /*[npair]; default mlim_spin0*/);
};

// Spin-2 Legendre tables: the two spin-weighted chains (+2, -2) share alpha / cnorm and differ by the sign of beta:
//   mu+-_{l+1} = (alpha_{l+1} x +- beta_{l+1}) mu+-_l - mu+-_{l-1},   (+-2)lambda_lm = cnorm_l mu+-_l,   l >= l0 = max(m, 2)
// from (pinned in the tests against the Goldberg closed form of the spin-weighted harmonics):
//   s_lam_{l+1} = [(x + s m/(l(l+1))) s_lam_l - C_l s_lam_{l-1}] / C_{l+1},  C_l = sqrt((l^2-m^2)(l^2-4)/(l^2(4l^2-1)))
struct Legendre2Tables {
    int lmax = -1, npair_pad = 0, R = 2, nchunk = 0;
    std::vector<int> mlim;               // [npair_pad] spin-2 cut
    std::vector<double> alpha, beta, cnorm;   // [ntrip]
    std::vector<int> ls;                 // [(lmax+1) * npair_pad]
    std::vector<double> seed;            // [(lmax+1) * npair_pad * 4]: mu+_ls, mu+_{ls-1}, mu-_ls, mu-_{ls-1}
    std::vector<WaveTask> tasks;         // R pairs per lane, 4 tasks per workgroup, longest first
    bool uniform_start = false;          // every 64-pair lane block switches on at one l == max(m, 2) (mod 32)
    std::vector<int> lw_chunk;           // [(lmax+1) * nchunk] first l the adjoint writes for (m, chunk); lmax+2: nothing
    void build(int lmax, const std::vector<double>& x, const std::vector<double>& sth, int npair_pad, int R,
               int nthreads = 0, const std::vector<int>* mlim_in = nullptr /*[npair]; default mlim_spin2*/);
};

inline int mlim_spin2(int lmax, double sth, double cth) {   // libsharp's cut with spin = 2
    double ofs = lmax * 0.01;
    if (ofs < 100.) ofs = 100.;
    const double b = -4.0 * std::fabs(cth);
    const double t1 = lmax * sth + ofs;
    const double c = 4.0 - t1 * t1;
    const double discr = b * b - 4 * c;
    if (discr <= 0) return lmax;
    double res = (-b + std::sqrt(discr)) / 2.;
    if (res > lmax) res = lmax;
    return (int)(res + 0.5);
}

struct RingPairDesc {   // one north/south ring pair (or the equator alone: startS = -1)
    int nphi;           // pixels per ring
    int log2M;          // FFT size class: nt if power of two, else Bluestein M >= 2 nt - 1 (nt = nphi, or nphi/2 if split)
    int bluestein;      // 0/1
    int split;          // 0, or 1 + scratch line index: ring done as two half-length transforms (kernels_body.hpp)
    int mmax_eff;       // = mlim of the pair
    int64_t startN, startS;  // offsets of the two rings in the *local* map (startS = -1: no southern ring)
    double phi0;
    double wgt;         // analysis weight W_ring * 4 pi / Npix
    int64_t chirp_off;  // offset into chirp table (Bluestein only)
    int ring;           // northern ring number 1..2*nside
    int log2T;          // 0, or log2 of the circulant size M_T >= 4 mmax_eff + 1 of the pair's Toeplitz form (mode 2)
    int64_t that_off;   // offset of the pair's multiplier spectrum (complex units) in a per-map array of that_elems
};

struct RingTables {
    int nside = 0, lmax = -1, npair = 0;
    int64_t npix_local = 0;
    std::vector<RingPairDesc> pairs;          // [npair]
    std::vector<std::vector<int>> classes;    // classes[log2M] = pair indices
    // mode 2 with multiplier spectra (kernels_body.hpp, Toeplitz form): pairs classed by the LDS image they then need;
    // classes_tb = the Toeplitz pairs by their Bluestein class (setup: t_d comes from a mode-1 transform)
    std::vector<std::vector<int>> classes_t, classes_tb;
    int64_t that_elems = 0;
    int log2Mmax = 0;
    int nsplit = 0, split_line = 0;           // number of split pairs, complex elements per scratch line (max n/2)
    std::vector<double> twiddle;              // [2 * Mmax/2]  exp(2 pi i k / Mmax), k < Mmax/2 (re,im)
    // Bluestein tables, per distinct nphi: w_j = exp(i pi j^2 / n) (j<n), then chat in bit-reversed order (M)
    std::vector<double> chirp;                // (re,im) pairs
    void build(int nside, int lmax, const std::vector<int>& rings /*northern ring numbers, ascending*/,
               const double* wring /*[2*nside] or nullptr*/, const std::vector<int>& mlim);
};

struct ShtTables {
    int nside = 0, lmax = -1;
    LegendreTables leg;
    Legendre2Tables leg2;     // built on demand (polarised plans)
    void build_spin2(int nthreads = 0);
    RingTables ring;
    // rings: northern ring numbers (1..2*nside) this plan owns; empty = all (single GPU)
    // max_maps: how many maps one call transforms at once; picks the ring pairs per lane R (more maps per wave ->
    // fewer ring pairs per lane, same register budget)
    void build(int nside, int lmax, const std::vector<int>& rings, const double* wring, int max_maps = 1,
               bool pol = false, int nthreads = 0);
};

// Gauss-Legendre nodes/weights on (-1,1), descending x (north first).
void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w);

}  // namespace cmdr
