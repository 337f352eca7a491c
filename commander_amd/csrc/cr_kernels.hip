// gfx950 kernels of the CR solver that are pure HBM streams:
// harmonic-space scalings / re-layouts on (l, m) grids, pixel-space noise weighting, fused CG vector updates and
// deterministic two-stage dot products whose results stay on the device (no host sync inside a CG iteration).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cr_body.hpp"
#include "kernels.hpp"

namespace cmdr {

// ----------------------------------------------------------------------------- (l, m)-grid kernels
// thread = one real slot of the packed a_lm row of m (the (+m, -m) pairs are interleaved): consecutive lanes touch
// consecutive doubles of every Stokes column
__global__ void k_sqrtS(const CompDev* __restrict__ comps, const double* __restrict__ smat, int kind,
                        const double* __restrict__ in, const double* __restrict__ add, double* __restrict__ out,
                        int pass_inactive) {
    const CompDev C = comps[blockIdx.z];
    const int m = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    if (m > C.lmax) return;
    const int l = m == 0 ? e : m + (e >> 1);
    if (l > C.lmax) return;
    const int64_t i = d_packed_index(C.lmax, l, m) + (m == 0 ? 0 : (e & 1));
    sqrtS_slot(C, smat, kind, in, add, out, l, i, pass_inactive != 0);
}
void launch_sqrtS(const CompDev* comps, int ncomp, int lmax_max, const double* smat, int kind, const double* in,
                  const double* add, double* out, bool pass_inactive, hipStream_t s) {
    dim3 grid((2 * (lmax_max + 1) + 255) / 256, lmax_max + 1, ncomp);
    hipLaunchKernelGGL(k_sqrtS, grid, dim3(256), 0, s, comps, smat, kind, in, add, out, pass_inactive ? 1 : 0);
}

// thread = (l, bm) with bm fastest: the nbm entries of a stream row are contiguous (144 B at 9 maps), so a wave
// writes one contiguous run; the sx value of an l is read once and broadcast to its nbm lanes
__global__ void k_band_prep(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ sx,
                            const double* __restrict__ w, const int* __restrict__ bm_stokes,
                            double* __restrict__ ast, int nbm, int lpb, const double* __restrict__ cnorm, int lmax_g,
                            const double* __restrict__ extra) {
    const int ll = threadIdx.x / nbm, bm = threadIdx.x - ll * nbm;
    if (ll >= lpb) return;
    const int m = blockIdx.y, l = m + blockIdx.x * lpb + ll;
    if (l > lmax_g + 1) return;
    const int64_t na = (int64_t)(lmax_g + 1) * (lmax_g + 1);
    band_prep_elem(comps, ncomp, sx, w + (int64_t)bm * ncomp * (lmax_g + 1), bm_stokes[bm], ast, nbm, bm, cnorm,
                   lmax_g, m, l, extra ? extra + bm * na : nullptr);
}
void launch_band_prep(const CompDev* comps, int ncomp, const double* sx, const double* w, const int* bm_stokes,
                      double* ast, const double* cnorm, int lmax_g, int nbm, hipStream_t s, const double* extra) {
    const int lpb = std::max(1, 256 / nbm);   // l values per block
    dim3 grid((lmax_g + 2 + lpb - 1) / lpb, lmax_g + 1);
    hipLaunchKernelGGL(k_band_prep, grid, dim3(256), 0, s, comps, ncomp, sx, w, bm_stokes, ast, nbm, lpb, cnorm, lmax_g,
                       extra);
}

__global__ void k_band_post(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ part,
                            int64_t pms, int64_t pcs, int nchunk, int nbm, const int* __restrict__ bm_stokes,
                            const double* __restrict__ w, const double* __restrict__ cnorm, int lmax_g, int lmax_max,
                            double* __restrict__ yc, int accumulate, const int* __restrict__ lwtab, int m0) {
    const int m = m0 + blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax_max) return;
    band_post_elem(comps, ncomp, part, pms, pcs, nchunk, nbm, bm_stokes, w, cnorm, lmax_g, yc, accumulate, m, l,
                   m <= lmax_g ? lwtab : nullptr);
}
void launch_band_post(const CompDev* comps, int ncomp, int lmax_max, const double* part, int64_t pms, int64_t pcs,
                      int nchunk, int nbm, const int* bm_stokes, const double* w, const double* cnorm, int lmax_g,
                      double* yc, bool accumulate, hipStream_t s, const int* lwtab, int m0, int m1) {
    if (m1 < 0 || m1 > lmax_max + 1) m1 = lmax_max + 1;     // columns m0 <= m < m1 (default: all)
    if (m1 <= m0) return;
    dim3 grid((lmax_max + 1 + 255) / 256, m1 - m0);
    hipLaunchKernelGGL(k_band_post, grid, dim3(256), 0, s, comps, ncomp, part, pms, pcs, nchunk, nbm, bm_stokes, w,
                       cnorm, lmax_g, lmax_max, yc, accumulate ? 1 : 0, lwtab, m0);
}

__global__ void k_band_prep2(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ sx,
                             const double* __restrict__ w, int nT, double* __restrict__ st, int npol,
                             const double* __restrict__ cnorm2, int lmax_g, const double* __restrict__ extra) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax_g + 1) return;
    const int ip = blockIdx.z;
    const int64_t ws = (int64_t)ncomp * (lmax_g + 1);
    const int64_t na = (int64_t)(lmax_g + 1) * (lmax_g + 1);
    band_prep2_elem(comps, ncomp, sx, w + (nT + 2 * ip) * ws, w + (nT + 2 * ip + 1) * ws, st, npol, ip, cnorm2, lmax_g,
                    m, l, extra ? extra + (nT + 2 * ip) * na : nullptr, extra ? extra + (nT + 2 * ip + 1) * na : nullptr);
}
void launch_band_prep2(const CompDev* comps, int ncomp, const double* sx, const double* w, int nT, double* st, int npol,
                       const double* cnorm2, int lmax_g, hipStream_t s, const double* extra) {
    dim3 grid((lmax_g + 2 + 255) / 256, lmax_g + 1, npol);
    hipLaunchKernelGGL(k_band_prep2, grid, dim3(256), 0, s, comps, ncomp, sx, w, nT, st, npol, cnorm2, lmax_g, extra);
}
__global__ void k_band_post2(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ part2,
                             int64_t pps, int64_t pcs, int nchunk, int npol, const double* __restrict__ w, int nT,
                             const double* __restrict__ cnorm2, int lmax_g, double* __restrict__ yc,
                             const int* __restrict__ lwtab) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax_g) return;
    band_post2_elem(comps, ncomp, part2, pps, pcs, nchunk, npol, w, nT, cnorm2, lmax_g, yc, m, l, lwtab);
}
void launch_band_post2(const CompDev* comps, int ncomp, int /*lmax_max*/, const double* part2, int64_t pps, int64_t pcs,
                       int nchunk, int npol, const double* w, int nT, const double* cnorm2, int lmax_g, double* yc,
                       hipStream_t s, const int* lwtab) {
    dim3 grid((lmax_g + 1 + 255) / 256, lmax_g + 1);
    hipLaunchKernelGGL(k_band_post2, grid, dim3(256), 0, s, comps, ncomp, part2, pps, pcs, nchunk, npol, w, nT, cnorm2,
                       lmax_g, yc, lwtab);
}

__global__ void k_alm_copy(const double* __restrict__ src, int lmax_s, double* __restrict__ dst, int lmax_d,
                           const double* __restrict__ fl, int accumulate, int lcut) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax_d) return;
    alm_copy_elem(src, lmax_s, dst, lmax_d, fl, accumulate, lcut, m, l);
}
void launch_alm_copy(const double* src, int lmax_s, double* dst, int lmax_d, const double* fl, bool accumulate,
                     hipStream_t s, int lcut) {
    dim3 grid((lmax_d + 1 + 255) / 256, lmax_d + 1);
    hipLaunchKernelGGL(k_alm_copy, grid, dim3(256), 0, s, src, lmax_s, dst, lmax_d, fl, accumulate ? 1 : 0, lcut);
}

// several columns in one launch (blockIdx.z): the staging copies of the varying-mixing batches, one per band
__global__ void k_alm_copy_batch(AlmCopyBatch B) {
    const AlmCopyDesc D = B.d[blockIdx.z];
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (m > D.lmax_d || l > D.lmax_d) return;
    alm_copy_elem(D.src, D.lmax_s, D.dst, D.lmax_d, D.fl, D.accumulate, D.lcut, m, l);
}
void launch_alm_copy_batch(const AlmCopyDesc* d, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += kAlmCopyBatch) {
        const int nb = std::min(kAlmCopyBatch, n - i0);
        AlmCopyBatch B;
        int lm = 0;
        for (int i = 0; i < nb; ++i) { B.d[i] = d[i0 + i]; lm = std::max(lm, d[i0 + i].lmax_d); }
        for (int i = nb; i < kAlmCopyBatch; ++i) B.d[i] = d[i0];
        dim3 grid((lm + 1 + 255) / 256, lm + 1, nb);
        hipLaunchKernelGGL(k_alm_copy_batch, grid, dim3(256), 0, s, B);
    }
}

__global__ void k_pinv_prior(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ Q, int lmax_pre,
                             int nmaps_pre, const double* __restrict__ x, const double* __restrict__ z,
                             double* __restrict__ out) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax_pre) return;
    pinv_prior_elem(comps, ncomp, Q, lmax_pre, nmaps_pre, x, z, out, m, l);
}
void launch_pinv_prior(const CompDev* comps, int ncomp, int /*lmax_max*/, const double* Q, int lmax_pre, int nmaps_pre,
                       const double* x, const double* z, double* out, hipStream_t s) {
    dim3 grid((lmax_pre + 1 + 255) / 256, lmax_pre + 1);
    hipLaunchKernelGGL(k_pinv_prior, grid, dim3(256), 0, s, comps, ncomp, Q, lmax_pre, nmaps_pre, x, z, out);
}

__global__ void k_precond_diag(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ P,
                               int lmax_pre, int nmaps_pre, const double* __restrict__ in,
                               double* __restrict__ out) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax_pre) return;
    precond_diag_elem(comps, ncomp, P, lmax_pre, nmaps_pre, in, out, m, l);
}
void launch_precond_diag(const CompDev* comps, int ncomp, const double* P, int lmax_pre, int nmaps_pre,
                         const double* in, double* out, hipStream_t s) {
    dim3 grid((lmax_pre + 1 + 255) / 256, lmax_pre + 1);
    hipLaunchKernelGGL(k_precond_diag, grid, dim3(256), 0, s, comps, ncomp, P, lmax_pre, nmaps_pre, in, out);
}

// fill the "phases" of a Gauss-Legendre Legendre plan with per-node weights: ph[m][pair] = (wN, 0, wS, 0)
__global__ void k_fill_gl(double* __restrict__ ph, const double* __restrict__ wn, const double* __restrict__ ws,
                          int npair_pad, int lmax) {
    const int p = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (p >= npair_pad) return;
    double* o = ph + d_phidx(lmax + 1, p, m);
    o[0] = wn[p];
    o[1] = 0.0;
    o[2] = ws[p];
    o[3] = 0.0;
}
void launch_fill_gl(double* ph, const double* wn, const double* ws, int npair_pad, int lmax, hipStream_t s) {
    dim3 grid((npair_pad + 255) / 256, lmax + 1);
    hipLaunchKernelGGL(k_fill_gl, grid, dim3(256), 0, s, ph, wn, ws, npair_pad, lmax);
}

// invN_diag: out packed (+m,-m both) = cnorm^2 * sum_chunks Re part   (comm_N_mod.f90:180-186)
__global__ void k_part_to_diag(const double* __restrict__ part, int64_t pcs, int nchunk,
                               const double* __restrict__ cnorm, double* __restrict__ out, int lmax) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax) return;
    const int64_t t = d_moffp(lmax, m) + (l - m);
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[c * pcs + 2 * t];
    const double cn = cnorm[t];
    s *= cn * cn;
    const int64_t i = d_packed_index(lmax, l, m);
    out[i] = s;
    if (m > 0) out[i + 1] = s;
}
void launch_part_to_diag(const double* part, int64_t pcs, int nchunk, const double* cnorm, double* out, int lmax,
                         hipStream_t s) {
    dim3 grid((lmax + 1 + 255) / 256, lmax + 1);
    hipLaunchKernelGGL(k_part_to_diag, grid, dim3(256), 0, s, part, pcs, nchunk, cnorm, out, lmax);
}

// ----------------------------------------------------------------------------- pixel-space streams
// mode 0: out = a*b ; 1: out = a*(a*b + c) (RHS "sample": sqrtInvN, + xi, sqrtInvN; comm_cr_mod.f90:600-609)
__global__ void k_pix(int mode, const double* __restrict__ a, const double* __restrict__ b,
                      const double* __restrict__ c, double* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double av = a[i];
        out[i] = mode == 0 ? av * b[i] : av * (av * b[i] + c[i]);
    }
}
void launch_pix(int mode, const double* a, const double* b, const double* c, double* out, int64_t n,
                hipStream_t s) {
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_pix, dim3(nb), dim3(256), 0, s, mode, a, b, c, out, n);
}

__global__ void k_index_copy(const double* __restrict__ src, const int64_t* __restrict__ idx, double* __restrict__ dst,
                             int n, int scatter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (scatter) dst[idx[i]] = src[i];
    else dst[i] = src[idx[i]];
}
void launch_index_copy(const double* src, const int64_t* idx, double* dst, int n, bool scatter, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_index_copy, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, dst, n, scatter ? 1 : 0);
}

// ----------------------------------------------------------------------------- CG vector algebra
// Deterministic dot: fixed grid of kDotBlocks partial sums (wave shuffle -> LDS), then one block folds them in
// a fixed order.  Results land in a small device scalar array so the host never waits inside an iteration.
constexpr int kDotBlocks = 1024;

__device__ inline double block_sum_256(double v) {
    __shared__ double sm[4];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) k_dot_partial(const double* __restrict__ a, const double* __restrict__ b,
                                                     int64_t n, double* __restrict__ partial) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)kDotBlocks * 256) acc += a[i] * b[i];
    const double r = block_sum_256(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// scal[slot] = sum(partial); if shift: scal[slot+1] = old scal[slot] first (delta_old <- delta_new)
__global__ void __launch_bounds__(256) k_dot_final(const double* __restrict__ partial, double* __restrict__ scal,
                                                   int slot, int shift) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < kDotBlocks; i += 256) acc += partial[i];
    const double r = block_sum_256(acc);
    if (threadIdx.x == 0) {
        if (shift) scal[slot + 1] = scal[slot];
        scal[slot] = r;
    }
}
void launch_dot(const double* a, const double* b, int64_t n, double* partial, double* scal, int slot, bool shift,
                hipStream_t s) {
    hipLaunchKernelGGL(k_dot_partial, dim3(kDotBlocks), dim3(256), 0, s, a, b, n, partial);
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(256), 0, s, partial, scal, slot, shift ? 1 : 0);
}
int dot_partial_count() { return kDotBlocks; }

// ---- fused PCG vector kernels (diagonal preconditioner, diffuse components only): 3 launches per iteration instead
// of 10.  Fixed grid of kDotBlocks blocks striding over the (m, l - m) rectangle; the dot products leave kDotBlocks
// partial sums that every block of the CONSUMING kernel folds itself, in one fixed order (same value in every block,
// no atomics, no separate fold launch).
__device__ inline double fold_partials(const double* __restrict__ partial) {
    __shared__ double tot;
    double acc = 0.0;
    for (int i = threadIdx.x; i < kDotBlocks; i += 256) acc += partial[i];
    const double r = block_sum_256(acc);
    if (threadIdx.x == 0) tot = r;
    __syncthreads();
    return tot;
}
// Block b walks the row pairs (m, lmax - m) with m = b, b + kDotBlocks, ...: a pair holds lmax + 2 entries whatever m,
// so the blocks are balanced without an index division; consecutive threads take consecutive l of one row (the (re, im)
// slots of consecutive l are adjacent in the packed layout: coalesced).
template <class F>
__device__ inline void cg_stride(int lmax, F f) {
    for (int m = blockIdx.x; 2 * m <= lmax; m += kDotBlocks) {
        for (int l = m + threadIdx.x; l <= lmax; l += 256) f(m, l);
        const int m2 = lmax - m;
        if (m2 != m)
            for (int l = m2 + threadIdx.x; l <= lmax; l += 256) f(m2, l);
    }
}
__global__ void __launch_bounds__(256) k_cg_q(const CompDev* __restrict__ comps, int ncomp, int lmax,
                                              const double* __restrict__ smat, const double* __restrict__ yc,
                                              const double* __restrict__ d, double* __restrict__ q,
                                              double* __restrict__ p_dq, int64_t ilo, int64_t ihi) {
    double acc = 0.0;
    if (cg_single(comps, ncomp)) {
        const CompDev C = comps[0];
        cg_stride(lmax, [&](int m, int l) { acc += cg_q_elem1(C, smat, yc, d, q, m, l, ilo, ihi); });
    } else {
        cg_stride(lmax, [&](int m, int l) { acc += cg_q_elem(comps, ncomp, smat, yc, d, q, m, l); });
    }
    const double r = block_sum_256(acc);
    if (threadIdx.x == 0) p_dq[blockIdx.x] = r;
}
__global__ void __launch_bounds__(256) k_cg_xr_precond(const CompDev* __restrict__ comps, int ncomp, int lmax,
                                                       const double* __restrict__ P, int nmaps_pre,
                                                       const double* __restrict__ p_dq,
                                                       const double* __restrict__ p_rs_old, double* __restrict__ p_rs,
                                                       double* __restrict__ x, double* r, const double* __restrict__ d,
                                                       const double* __restrict__ q, double* s,
                                                       double* __restrict__ scal, int64_t ilo, int64_t ihi) {
    const double dq = fold_partials(p_dq);
    const double alpha = fold_partials(p_rs_old) / dq;                                  // comm_cr_mod.f90:254
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[2] = dq;
    double acc = 0.0;
    if (cg_single(comps, ncomp) && nmaps_pre == 1) {
        const CompDev C = comps[0];
        cg_stride(lmax, [&](int m, int l) { acc += cg_xr_elem1(C, P, lmax, alpha, x, r, d, q, s, m, l, ilo, ihi); });
    } else {
        cg_stride(lmax, [&](int m, int l) { acc += cg_xr_elem(comps, ncomp, P, lmax, nmaps_pre, alpha, x, r, d, q, s, m, l); });
    }
    const double t = block_sum_256(acc);
    if (threadIdx.x == 0) p_rs[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256) k_cg_d_sqrtS(const CompDev* __restrict__ comps, int ncomp, int lmax,
                                                    const double* __restrict__ smat,
                                                    const double* __restrict__ p_rs_old,
                                                    const double* __restrict__ p_rs, double* d,
                                                    const double* __restrict__ s, double* __restrict__ sx,
                                                    double* __restrict__ scal, int64_t ilo, int64_t ihi) {
    const double dold = fold_partials(p_rs_old), dnew = fold_partials(p_rs);
    const double beta = dnew / dold;                                                    // :270-271
    if (blockIdx.x == 0 && threadIdx.x == 0) { scal[0] = dnew; scal[1] = dold; }
    if (cg_single(comps, ncomp)) {
        const CompDev C = comps[0];
        cg_stride(lmax, [&](int m, int l) { cg_d_elem1(C, smat, beta, d, s, sx, m, l, ilo, ihi); });
    } else {
        cg_stride(lmax, [&](int m, int l) { cg_d_elem(comps, ncomp, smat, beta, d, s, sx, m, l); });
    }
}
// p[0] = scal[slot], p[1..] = 0: a dot product computed by launch_dot enters the partial-sum protocol
__global__ void k_cg_seed(const double* __restrict__ scal, int slot, double* __restrict__ p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < kDotBlocks) p[i] = i == 0 ? scal[slot] : 0.0;
}
void launch_cg_seed(const double* scal, int slot, double* p, hipStream_t s) {
    hipLaunchKernelGGL(k_cg_seed, dim3(kDotBlocks / 256), dim3(256), 0, s, scal, slot, p);
}
void launch_cg_q(const CompDev* comps, int ncomp, int lmax, const double* smat, const double* yc, const double* d,
                 double* q, double* p_dq, hipStream_t s, int64_t ilo, int64_t ihi) {
    hipLaunchKernelGGL(k_cg_q, dim3(kDotBlocks), dim3(256), 0, s, comps, ncomp, lmax, smat, yc, d, q, p_dq, ilo, ihi);
}
void launch_cg_xr_precond(const CompDev* comps, int ncomp, int lmax, const double* P, int nmaps_pre, const double* p_dq,
                          const double* p_rs_old, double* p_rs, double* x, double* r, const double* d, const double* q,
                          double* sv, double* scal, hipStream_t s, int64_t ilo, int64_t ihi) {
    hipLaunchKernelGGL(k_cg_xr_precond, dim3(kDotBlocks), dim3(256), 0, s, comps, ncomp, lmax, P, nmaps_pre, p_dq,
                       p_rs_old, p_rs, x, r, d, q, sv, scal, ilo, ihi);
}
void launch_cg_d_sqrtS(const CompDev* comps, int ncomp, int lmax, const double* smat, const double* p_rs_old,
                       const double* p_rs, double* d, const double* sv, double* sx, double* scal, hipStream_t s,
                       int64_t ilo, int64_t ihi) {
    hipLaunchKernelGGL(k_cg_d_sqrtS, dim3(kDotBlocks), dim3(256), 0, s, comps, ncomp, lmax, smat, p_rs_old, p_rs, d, sv, sx,
                       scal, ilo, ihi);
}

// x += alpha d ; r -= alpha q ; alpha = scal[num] / scal[den]   (comm_cr_mod.f90:254-261)
__global__ void k_cg_xr(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ d,
                        const double* __restrict__ q, int64_t n, const double* __restrict__ scal, int num, int den) {
    const double alpha = scal[num] / scal[den];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        x[i] += alpha * d[i];
        r[i] -= alpha * q[i];
    }
}
void launch_cg_xr(double* x, double* r, const double* d, const double* q, int64_t n, const double* scal, int num,
                  int den, hipStream_t s) {
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_cg_xr, dim3(nb), dim3(256), 0, s, x, r, d, q, n, scal, num, den);
}

// d = s + beta d ; beta = scal[num] / scal[den]   (comm_cr_mod.f90:271-272)
__global__ void k_cg_d(double* __restrict__ d, const double* __restrict__ sv, int64_t n,
                       const double* __restrict__ scal, int num, int den) {
    const double beta = scal[num] / scal[den];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = sv[i] + beta * d[i];
}
void launch_cg_d(double* d, const double* sv, int64_t n, const double* scal, int num, int den, hipStream_t s) {
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_cg_d, dim3(nb), dim3(256), 0, s, d, sv, n, scal, num, den);
}

// out = a + cb * b  (cb = +-1 etc.)
__global__ void k_axpby(const double* __restrict__ a, const double* __restrict__ b, double cb,
                        double* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a[i] + cb * b[i];
}
void launch_axpby(const double* a, const double* b, double cb, double* out, int64_t n, hipStream_t s) {
    const int nb = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_axpby, dim3(nb), dim3(256), 0, s, a, b, cb, out, n);
}

// ----------------------------------------------------------------------------- sigma_l (K13)
__global__ void k_compact_fwd(double* __restrict__ z, CellBase B, const int64_t* __restrict__ rows,
                              const int64_t* __restrict__ ptr, const int* __restrict__ col,
                              const double* __restrict__ val, const double* __restrict__ a, int64_t nrows) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < nrows) compact_fwd_row(z, B, rows, ptr, col, val, a, r);
}
void launch_compact_fwd(double* z, const CellBase& B, const int64_t* rows, const int64_t* ptr, const int* col,
                        const double* val, const double* a, int64_t nrows, hipStream_t s) {
    if (nrows == 0) return;
    hipLaunchKernelGGL(k_compact_fwd, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, z, B, rows, ptr, col, val, a,
                       nrows);
}
// kCadjSlices workgroups per parameter, each over a contiguous slice of the column (dense template columns have
// 10^7 entries), then one pass that adds the slice sums in a fixed order (deterministic)
constexpr int kCadjSlices = 64;
__global__ void __launch_bounds__(256) k_compact_adj(const double* __restrict__ u, CellBase B,
                                                     const int64_t* __restrict__ cptr, const int64_t* __restrict__ cell,
                                                     const double* __restrict__ val, double* __restrict__ partial) {
    const int p = blockIdx.x, sl = blockIdx.y;
    const int64_t k0 = cptr[p], len = cptr[p + 1] - k0;
    const int64_t chunk = (len + kCadjSlices - 1) / kCadjSlices;
    const int64_t lo = k0 + sl * chunk, hi = min(k0 + len, lo + chunk);
    double acc = 0.0;
    for (int64_t k = lo + threadIdx.x; k < hi; k += 256) acc += compact_adj_term(u, B, cell, val, k);
    const double r = block_sum_256(acc);
    if (threadIdx.x == 0) partial[(int64_t)p * kCadjSlices + sl] = r;
}
__global__ void k_compact_adj_final(const double* __restrict__ partial, const double* __restrict__ scale,
                                    double* __restrict__ y, int nparam, int accumulate) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nparam) return;
    double r = 0.0;
    for (int sl = 0; sl < kCadjSlices; ++sl) r += partial[(int64_t)p * kCadjSlices + sl];
    const double v = r * (scale ? scale[p] : 1.0);
    y[p] = accumulate ? y[p] + v : v;
}
void launch_compact_adj(const double* u, const CellBase& B, const int64_t* cptr, const int64_t* cell, const double* val,
                        const double* scale, double* y, int nparam, bool accumulate, double* scratch, hipStream_t s) {
    if (nparam == 0) return;
    hipLaunchKernelGGL(k_compact_adj, dim3(nparam, kCadjSlices), dim3(256), 0, s, u, B, cptr, cell, val, scratch);
    hipLaunchKernelGGL(k_compact_adj_final, dim3((nparam + 255) / 256), dim3(256), 0, s, scratch, scale, y, nparam,
                       accumulate ? 1 : 0);
}
int compact_adj_scratch(int nparam) { return nparam * kCadjSlices; }
__global__ void __launch_bounds__(256) k_dense_mv(const double* __restrict__ M, const double* __restrict__ x,
                                                  double* __restrict__ y, int n) {
    const int i = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) acc += M[(int64_t)i * n + j] * x[j];
    const double r = block_sum_256(acc);
    if (threadIdx.x == 0) y[i] = r;
}
void launch_dense_mv(const double* M, const double* x, double* y, int n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_dense_mv, dim3(n), dim3(256), 0, s, M, x, y, n);
}
__global__ void k_vec_scale(int mode, const double* __restrict__ a, const double* __restrict__ sc,
                            const double* __restrict__ b, const double* __restrict__ c, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = mode == 1 ? a[i] / sc[i] : a[i] * sc[i];
    if (mode >= 2 && b) v += b[i];
    if (mode == 3 && c) v += c[i] / sc[i];
    out[i] = v;
}
void launch_vec_scale(int mode, const double* a, const double* sc, const double* b, const double* c, double* out, int n,
                      hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_vec_scale, dim3((n + 255) / 256), dim3(256), 0, s, mode, a, sc, b, c, out, n);
}

// Phase-array slots of maps that share a synthesis (bcast: slot j <- slot 0) or whose adjoints are summed into one
// column (sum: slot 0 += slots 1..n-1, fixed order); slot j starts at base + j * slot_stride.
__global__ void __launch_bounds__(256) k_phase_share(double* __restrict__ base, int64_t slot_stride, int n, int64_t elems,
                                                     int sum) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= elems) return;
    double2* p0 = reinterpret_cast<double2*>(base + i);
    double2 v = *p0;
    if (sum) {
        for (int j = 1; j < n; ++j) {
            const double2 w = *reinterpret_cast<const double2*>(base + j * slot_stride + i);
            v.x += w.x;
            v.y += w.y;
        }
        *p0 = v;
    } else {
        for (int j = 1; j < n; ++j) *reinterpret_cast<double2*>(base + j * slot_stride + i) = v;
    }
}
void launch_phase_share(double* base, int64_t slot_stride, int n, int64_t elems, bool sum, hipStream_t s) {
    if (n <= 1 || elems <= 0) return;
    hipLaunchKernelGGL(k_phase_share, dim3((unsigned)((elems / 2 + 255) / 256)), dim3(256), 0, s, base, slot_stride, n, elems,
                       sum ? 1 : 0);
}

__global__ void k_alm_chain(double* __restrict__ alm, int64_t alm_stride, float* __restrict__ c32, int lmax,
                            int to_chain) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax) return;
    const int64_t na = (int64_t)(lmax + 1) * (lmax + 1);
    alm_chain_elem(alm + blockIdx.z * alm_stride, c32 + blockIdx.z * na, lmax, to_chain, m, l);
}
void launch_alm_chain(double* alm, int64_t alm_stride, float* c32, int lmax, int nmaps, bool to_chain, hipStream_t s) {
    dim3 grid((lmax + 1 + 255) / 256, lmax + 1, nmaps);
    hipLaunchKernelGGL(k_alm_chain, grid, dim3(256), 0, s, alm, alm_stride, c32, lmax, to_chain ? 1 : 0);
}

// getSigmaL (commander3/src/comm_map_mod.f90:1302-1351): sigma_l(l, k) = sum_{m=-l..l} a_lm^i a_lm^j / (2l+1) for
// the nspec = nmaps(nmaps+1)/2 pairs (i<=j) in Commander's order.  One workgroup per l; fixed-order block reduction.
__global__ void __launch_bounds__(256) k_sigma_l(const double* __restrict__ alm, int64_t stride, int lmax, int nmaps,
                                                 double* __restrict__ out) {
    const int l = blockIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int m = threadIdx.x; m <= l; m += 256) {
        const int64_t i0 = d_packed_index(lmax, l, m);
        for (int sl = 0; sl < (m == 0 ? 1 : 2); ++sl) {
            double v[3];
            for (int a = 0; a < nmaps; ++a) v[a] = alm[a * stride + i0 + sl];
            int k = 0;
            for (int a = 0; a < nmaps; ++a)
                for (int b = a; b < nmaps; ++b) acc[k++] += v[a] * v[b];
        }
    }
    const int nspec = nmaps * (nmaps + 1) / 2;
    for (int k = 0; k < nspec; ++k) {
        const double r = block_sum_256(acc[k]);
        if (threadIdx.x == 0) out[l + (int64_t)(lmax + 1) * k] = r / (double)(2 * l + 1);
    }
}
void launch_sigma_l(const double* alm, int64_t stride, int lmax, int nmaps, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sigma_l, dim3(lmax + 1), dim3(256), 0, s, alm, stride, lmax, nmaps, out);
}

}  // namespace cmdr
