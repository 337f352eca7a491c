// Launchers of the gfx950 kernels (implemented in kernels.hip / cr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>

#include "cr_body.hpp"
#include "kernels_body.hpp"
#include "plan_tables.hpp"

namespace cmdr {

// coefficient stream: maps interleaved, ast[((t * nmaps) + k) * 2 + {re, im}]
int leg_max_batch(int R);
// nbs: maps interleaved in the stream buffer (default nmaps); a sub-range of maps is addressed by shifting ast / ph
// prep != null (only when leg_synth_can_prep(A)): the coefficients are formed from the stacked vector while the kernel
// stages its tiles (k_band_prep folded in) and ast is not read
void launch_leg_synth(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, double* ph,
                      int64_t ph_stride, int nmaps, hipStream_t s, int nbs = -1, const PrepDev* prep = nullptr);
bool leg_synth_can_prep(const LegArgs& A);
// between(nmx): called once after the matrix-unit launches (nmx maps went through k_leg_adj_mx; 0 = none) and before
// the VALU launches of the remaining maps (profiling hook; may be empty)
void launch_leg_adj(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                    double* part, int64_t part_map_stride, int64_t part_chunk_stride, int nmaps, bool square,
                    hipStream_t s, const std::function<void(int)>& between = nullptr);
// mode 0: phases->map, 1: map->phases, 2: phases -> *mul -> phases (in place)
void launch_ring(int mode, const RingDev* rings, const int* cls, int ncls, int log2M, double* ph,
                 int64_t ph_stride, int64_t prow /* rows (m) per pair of the phase layout */, double* map, int64_t map_stride, const double* const* mul,
                 int weighted, const cd* tw, int log2Mmax, const cd* chirp, cd* scratch, int64_t scratch_map_stride,
                 int scratch_line, int nmaps, hipStream_t s, const cd* that = nullptr, int64_t that_stride = 0);
// multiplier spectra of the Toeplitz pairs in cls (class log2M = their circulant size) from t_d in phase layout
void launch_ring_toeplitz_spec(const RingDev* rings, const int* cls, int ncls, int log2M, const double* td,
                               int64_t prow, cd* that, const cd* tw, int log2Mmax, hipStream_t s);
// masked monopole / dipole sums per ring pair (applyMonoDipolePrior): out[npair][16], see md_pixel_accum
void launch_md_sums(const RingDev* rings, int npair, int nside, const double* map, const double* mask, int type,
                    double* out, hipStream_t s);
void launch_alm_to_stream(const double* alm, int64_t alm_stride, double* ast, const double* cnorm, int lmax,
                          int nmaps, hipStream_t s);
// lwtab (optional): [(lmax+1) * nchunk] first l written for (m, chunk) -- entries below it are structurally zero and
// are not read (a quarter of the partial columns at Nside 1024 / lmax 2000)
void launch_part_to_alm(const double* part, int64_t pms, int64_t pcs, int nchunk, double* alm, int64_t alm_stride,
                        const double* cnorm, int lmax, int nmaps, hipStream_t s, const int* lwtab = nullptr);

// ---- spin-2 Legendre stage: npol polarisation pairs; pair ip uses phase maps kq0 + 2 ip (Q) and kq0 + 2 ip + 1 (U);
// stream st[((t * npol) + ip) * 4 + {E'r,E'i,B'r,B'i}]; partials part[ip][chunk][4 * padded triangle]
void launch_leg2_synth(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* st, int npol, double* ph,
                       int64_t ph_stride, int kq0, hipStream_t s);
void launch_leg2_adj(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                     int kq0, double* part, int64_t part_pol_stride, int64_t part_chunk_stride, int npol,
                     hipStream_t s);
void launch_alm2_to_stream(const double* aE, const double* aB, int64_t pol_stride, double* st, int npol,
                           const double* cnorm, int lmax, hipStream_t s);
void launch_part2_to_alm(const double* part, int64_t part_pol_stride, int64_t pcs, int nchunk, double* aE, double* aB,
                         int64_t pol_stride, const double* cnorm, int lmax, int npol, hipStream_t s, const int* lwtab = nullptr);

// ---- CR solver streams (cr_kernels.hip)
void launch_sqrtS(const CompDev* comps, int ncomp, int lmax_max, const double* smat, int kind, const double* in,
                  const double* add, double* out, bool pass_inactive, hipStream_t s);
// extra (optional): [nbm] packed a_lm(lmax_g) columns added to the band signal (varying-mixing components)
void launch_band_prep(const CompDev* comps, int ncomp, const double* sx, const double* w, const int* bm_stokes,
                      double* ast, const double* cnorm, int lmax_g, int nbm, hipStream_t s,
                      const double* extra = nullptr);
void launch_band_post(const CompDev* comps, int ncomp, int lmax_max, const double* part, int64_t pms, int64_t pcs,
                      int nchunk, int nbm, const int* bm_stokes, const double* w, const double* cnorm, int lmax_g,
                      double* yc, bool accumulate, hipStream_t s, const int* lwtab = nullptr, int m0 = 0, int m1 = -1);
void launch_band_prep2(const CompDev* comps, int ncomp, const double* sx, const double* w, int nT, double* st, int npol,
                       const double* cnorm2, int lmax_g, hipStream_t s, const double* extra = nullptr);
void launch_band_post2(const CompDev* comps, int ncomp, int lmax_max, const double* part2, int64_t pps, int64_t pcs,
                       int nchunk, int npol, const double* w, int nT, const double* cnorm2, int lmax_g, double* yc,
                       hipStream_t s, const int* lwtab = nullptr);
void launch_precond_diag(const CompDev* comps, int ncomp, const double* P, int lmax_pre, int nmaps_pre,
                         const double* in, double* out, hipStream_t s);
void launch_alm_copy_batch(const AlmCopyDesc* d, int n, hipStream_t s);   // columns with distinct destinations
void launch_alm_copy(const double* src, int lmax_s, double* dst, int lmax_d, const double* fl, bool accumulate,
                     hipStream_t s, int lcut = 1 << 30);
void launch_pinv_prior(const CompDev* comps, int ncomp, int lmax_max, const double* Q, int lmax_pre, int nmaps_pre,
                       const double* x, const double* z, double* out, hipStream_t s);
void launch_fill_gl(double* ph, const double* wn, const double* ws, int npair_pad, int lmax, hipStream_t s);
void launch_part_to_diag(const double* part, int64_t pcs, int nchunk, const double* cnorm, double* out, int lmax,
                         hipStream_t s);
void launch_pix(int mode, const double* a, const double* b, const double* c, double* out, int64_t n, hipStream_t s);
int dot_partial_count();
void launch_dot(const double* a, const double* b, int64_t n, double* partial, double* scal, int slot, bool shift,
                hipStream_t s);
// fused PCG vector kernels (diagonal preconditioner, diffuse components only): p_* = dot_partial_count() partial sums
void launch_cg_seed(const double* scal, int slot, double* p, hipStream_t s);
void launch_cg_q(const CompDev* comps, int ncomp, int lmax, const double* smat, const double* yc, const double* d,
                 double* q, double* p_dq, hipStream_t s, int64_t ilo = 0, int64_t ihi = INT64_MAX);
void launch_cg_xr_precond(const CompDev* comps, int ncomp, int lmax, const double* P, int nmaps_pre, const double* p_dq,
                          const double* p_rs_old, double* p_rs, double* x, double* r, const double* d, const double* q,
                          double* sv, double* scal, hipStream_t s, int64_t ilo = 0, int64_t ihi = INT64_MAX);
void launch_cg_d_sqrtS(const CompDev* comps, int ncomp, int lmax, const double* smat, const double* p_rs_old,
                       const double* p_rs, double* d, const double* sv, double* sx, double* scal, hipStream_t s,
                       int64_t ilo = 0, int64_t ihi = INT64_MAX);
void launch_cg_xr(double* x, double* r, const double* d, const double* q, int64_t n, const double* scal, int num,
                  int den, hipStream_t s);
void launch_cg_d(double* d, const double* sv, int64_t n, const double* scal, int num, int den, hipStream_t s);
void launch_axpby(const double* a, const double* b, double cb, double* out, int64_t n, hipStream_t s);
// compact components: z += P a (CSR over the touched cells), y[p] (+)= scale[p] * (P^t u)[p] (CSC), small dense y = M x
void launch_compact_fwd(double* z, const CellBase& B, const int64_t* rows, const int64_t* ptr, const int* col,
                        const double* val, const double* a, int64_t nrows, hipStream_t s);
void launch_compact_adj(const double* u, const CellBase& B, const int64_t* cptr, const int64_t* cell, const double* val,
                        const double* scale, double* y, int nparam, bool accumulate, double* scratch, hipStream_t s);
int compact_adj_scratch(int nparam);   // doubles of scratch launch_compact_adj needs
void launch_dense_mv(const double* M, const double* x, double* y, int n, hipStream_t s);
// mode 0: out = a * s ; 1: out = a / s ; 2: out = a * s + b ; 3: out = a * s + b + c / s  (b, c nullable -> 0)
void launch_vec_scale(int mode, const double* a, const double* sc, const double* b, const double* c, double* out, int n,
                      hipStream_t s);
// dst[i] = src[idx[i]] (scatter = false) or dst[idx[i]] = src[i] (scatter = true), i < n: the low-l preconditioner's
// (l, m) <-> stacked-vector bookkeeping
void launch_index_copy(const double* src, const int64_t* idx, double* dst, int n, bool scatter, hipStream_t s);
void launch_phase_share(double* base, int64_t slot_stride, int n, int64_t elems, bool sum, hipStream_t s);
void launch_alm_chain(double* alm, int64_t alm_stride, float* c32, int lmax, int nmaps, bool to_chain, hipStream_t s);
void launch_sigma_l(const double* alm, int64_t stride, int lmax, int nmaps, double* out, hipStream_t s);

}  // namespace cmdr
