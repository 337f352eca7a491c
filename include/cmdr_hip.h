/*
 * cmdr_hip.h -- C ABI of libcmdr_hip.so: the MI355X-native replacement for the Gibbs amplitude-sampling path of
 * Commander3 (constrained-realization PCG solve + the spherical-harmonic transforms it calls).
 *
 * Two entry levels:
 *   (1) SHT level -- what commander3/src/sharp.f90 binds from libsharp2 today, with explicit sizes and an
 *       int status instead of void/abort (the literal libsharp2 symbol names are in include/cmdr_sharp.h);
 *   (2) CR level -- the bodies of cr_matmulA / cr_invM / cr_computeRHS / solve_cr_eqn_by_CG
 *       (commander3/src/comm_cr_mod.f90) with all vectors resident in HBM for the whole solve.
 *
 * Conventions: every function returns 0 on success, a negative value on error (message via cmdr_last_error());
 * all arrays are fp64 (Fortran real(dp) / c_double), column-major exactly as the Fortran side holds them;
 * "packed a_lm" is Commander's m-major real-packed layout (commander3/src/comm_map_mod.f90:228-261);
 * maps are HEALPix RING ordered, restricted to the rings the plan owns in ascending ring order
 * (commander3/src/comm_map_mod.f90:193-226).  One host thread per context; one context per (chain, GPU).
 * The library is HIP-only: it fails loudly (error code) when no gfx950 device is usable -- there is no CPU path.
 */
#ifndef CMDR_HIP_H
#define CMDR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmdr_sht_plan cmdr_sht_plan;
typedef struct cmdr_ctx cmdr_ctx;

/* job types: same values as commander3/src/sharp.f90:8-14 */
enum { CMDR_YtW = 0, CMDR_Y = 1, CMDR_Yt = 2, CMDR_WY = 3 };

const char* cmdr_last_error(void);
/* number of visible HIP devices (0 if none / runtime unusable) */
int cmdr_device_count(void);
int cmdr_set_device(int device);
int cmdr_device_synchronize(void);

/* ---- device memory helpers (so a host language without HIP bindings can keep data resident) ---- */
int cmdr_dev_alloc(size_t nbytes, void** out);
int cmdr_dev_free(void* p);
/* hipMemGetInfo of the current device: free and total bytes (sizing bands / plans against the 288 GB of one MI355X). */
int cmdr_dev_mem_info(size_t* free_bytes, size_t* total_bytes);
int cmdr_memcpy_h2d(void* dst_dev, const void* src_host, size_t nbytes);
int cmdr_memcpy_d2h(void* dst_host, const void* src_dev, size_t nbytes);
/* Page-lock a caller-owned host buffer (hipHostRegister), so that the host-pointer entry points below copy from / to it by
 * DMA at the link's rate instead of staging pageable memory: the band maps Commander allocates once per run
 * (data(i)%res%map and the noise draws of cr_computeRHS, comm_cr_mod.f90:452-466) are 1.8 GB per amplitude sample at the
 * benchmark size.  Optional; the buffer must stay allocated (and must not move) until cmdr_host_unregister. */
int cmdr_host_register(void* ptr_host, size_t nbytes);
int cmdr_host_unregister(void* ptr_host);

/* ------------------------------------------------------------------------------------------------
 * SHT level.  Replaces sharp_make_mmajor_real_packed_alm_info + sharp_make_subset_healpix_geom_info
 * (sharp.f90:44-71, called from comm_map_mod.f90:264-283) and sharp_execute (sharp.f90:86-104, called from
 * comm_map_mod.f90:437-579).
 *   rings   : nrings northern ring numbers (1..2*nside) owned by this plan; the mirror ring 4*nside-i is
 *             implied (comm_map_mod.f90:197-221).  NULL / 0 = all rings.
 *   wring   : [2*nside] ring weights W (Commander passes 1 + weight_ring); NULL = 1.
 *   max_maps: how many columns one call may transform at once (workspace is sized for it).
 */
int cmdr_sht_plan_create(int nside, int lmax, int nrings, const int* rings, const double* wring, int max_maps,
                         cmdr_sht_plan** out);
/* same, with the spin-2 tables for the polarisation columns (comm_mapinfo with pol = .true.,
 * comm_map_mod.f90:134-137, geom_info_P :279-282) */
int cmdr_sht_plan_create_pol(int nside, int lmax, int nrings, const int* rings, const double* wring, int max_maps,
                             cmdr_sht_plan** out);
int cmdr_sht_plan_destroy(cmdr_sht_plan* plan);
int64_t cmdr_sht_nalm(const cmdr_sht_plan* plan);   /* sharp_alm_count  (sharp.f90:52-56) */
int64_t cmdr_sht_npix(const cmdr_sht_plan* plan);   /* sharp_map_size   (sharp.f90:78-82) */
/* One spin-0 transform per column; alm[k] / map[k] are the column pointers, as in sharp_execute (sharp.f90:203-224).
 * Host-pointer form copies in and out; the _dev form takes device pointers and leaves results in HBM. */
int cmdr_sht_execute(cmdr_sht_plan* plan, int job, int nmaps, double* const* alm, double* const* map);
int cmdr_sht_execute_dev(cmdr_sht_plan* plan, int job, int nmaps, double* alm_dev, int64_t alm_stride,
                         double* map_dev, int64_t map_stride);
/* The spin-2 call Commander issues on columns 2:3, (Q,U) <-> (E,B) (comm_map_mod.f90:446-449, 519-523, 549-553):
 * alm = (E, B), map = (Q, U), HEALPix "COSMO" convention a_{+-2,lm} = -(E_lm +- i B_lm).  Needs a _pol plan. */
int cmdr_sht_execute_spin2(cmdr_sht_plan* plan, int job, double* almE, double* almB, double* mapQ, double* mapU);
int cmdr_sht_execute_spin2_dev(cmdr_sht_plan* plan, int job, double* almE_dev, double* almB_dev, double* mapQ_dev,
                               double* mapU_dev);

/* ------------------------------------------------------------------------------------------------
 * CR level: the constrained-realization system of commander3/src/comm_cr_mod.f90 for diffuse components with
 * constant (F_mean) or spatially varying (F map) mixing and white per-pixel noise; T and T,Q,U bands.  Call order:
 *   cmdr_ctx_create -> [cmdr_ctx_set_rings] -> cmdr_band_add (every band, in data(i) order) ->
 *   cmdr_comp_add (every diffuse component, in compList order) -> cmdr_finalize ->
 *   cmdr_precond_init_diag -> cmdr_precond_update_diag -> cmdr_compute_rhs / cmdr_solve / cmdr_matmulA / cmdr_invM
 * The stacked vector x(1:ncr) has exactly Commander's layout (comm_cr_mod.f90:467-501: components in list order,
 * Stokes outer, packed a_lm inner).  Functions without a _dev suffix take HOST pointers and copy; _dev variants
 * take DEVICE pointers (cmdr_dev_alloc) and leave every result in HBM.
 */
int cmdr_ctx_create(int device, cmdr_ctx** out);
int cmdr_ctx_destroy(cmdr_ctx* ctx);
/* Multi-GPU: restrict every band of this nside to the ring pairs led by these northern rings
 * (Commander's own distribution, comm_map_mod.f90:197-221).  a_lm vectors stay replicated on every rank. */
int cmdr_ctx_set_rings(cmdr_ctx* ctx, int nside, int nrings, const int* rings);
/* In-place sum over ranks of n doubles at a DEVICE address; supplied by the host language (MPI in the Fortran
 * driver, torch.distributed/RCCL in bench.py).  Called once per cr_matmulA / cr_computeRHS on the partial
 * sum_bands(...) vector, replacing libsharp2's MPI exchange + mpi_dot_product's allreduce
 * (comm_utils.f90:599-614). */
typedef void (*cmdr_allreduce_fn)(void* user, double* dev_ptr, int64_t n);
int cmdr_ctx_set_allreduce(cmdr_ctx* ctx, cmdr_allreduce_fn fn, void* user);
/* Stream-ordered variant for collective libraries that take a stream (RCCL): fn must ENQUEUE the in-place sum on
 * hip_stream (the library's own hipStream_t) and may return before it has run.  No host synchronisation happens per
 * matvec then, so a whole fixed_iter solve is queued ahead of the GPU.  Takes precedence over the blocking callback. */
typedef void (*cmdr_allreduce_stream_fn)(void* user, double* dev_ptr, int64_t n, void* hip_stream);
int cmdr_ctx_set_allreduce_stream(cmdr_ctx* ctx, cmdr_allreduce_stream_fn fn, void* user);
/* Band x ring-set hybrid sharding (both partitions at once): this rank holds a subset of the bands
 * (cmdr_band_add only those; F_mean rows likewise) on a subset of the rings.  rings_fn sums over the ranks that hold
 * the SAME bands (may be NULL when that group is one rank); the callback of cmdr_ctx_set_allreduce[_stream] sums over
 * all ranks.  ring_replicas = number of ranks per band subset.  Diagonal preconditioner only. */
int cmdr_ctx_set_band_sharding(cmdr_ctx* ctx, cmdr_allreduce_fn rings_fn, void* user, int ring_replicas);
/* RCCL inside the library (the MI355X-native form of the exchange; RCCL = backend "nccl" of torch.distributed, bound
 * here with dlopen("librccl.so.1"), no link-time dependency).  One rank calls cmdr_rccl_unique_id and the host language
 * broadcasts the 128 bytes (MPI_Bcast in the Fortran driver, exactly where it already broadcasts parameters;
 * torch.distributed in bench.py); every rank then calls cmdr_ctx_init_rccl (collective; ncclCommInitRank on the
 * context's device).  From then on the sum over ranks of cr_matmulA / cr_computeRHS / preconditioner setup --
 * what libsharp2's MPI exchange + mpi_dot_product's MPI_Allreduce (comm_utils.f90:599-614) do in the reference --
 * is one ncclAllReduce per call, enqueued on the library's own HIP stream: no callback, no host synchronisation, a
 * whole fixed_iter solve stays queued ahead of the GPU.  Takes precedence over the callbacks above.
 * cmdr_ctx_rccl_split_rings: band x ring-set hybrid without callbacks -- ncclCommSplit(color = band_group,
 * key = ring_index) gives the communicator of the ranks holding the same bands; collective over ALL ranks (ranks
 * outside any hybrid layout do not call it).  Replaces cmdr_ctx_set_band_sharding.
 * cmdr_ctx_drop_rccl: destroys the context's communicators (and the band sharding a split set up), so that the callbacks
 * of cmdr_ctx_set_allreduce[_stream] / cmdr_ctx_set_band_sharding apply again -- for a driver whose ranks did not ALL get
 * a communicator and fall back together.  No-op without one.
 * cmdr_ctx_rccl_size: ncclCommCount read back from the communicator (0 = none).  cmdr_rccl_version: ncclGetVersion
 * (< 0: librccl could not be loaded; cmdr_last_error says why). */
int cmdr_rccl_unique_id(char* out128);
int cmdr_rccl_version(void);
int cmdr_ctx_init_rccl(cmdr_ctx* ctx, const char* id128, int rank, int nranks);
int cmdr_ctx_rccl_split_rings(cmdr_ctx* ctx, int band_group, int ring_index, int ring_replicas);
int cmdr_ctx_drop_rccl(cmdr_ctx* ctx);
/* m-sliced CG vectors (optional; the reference's ownership of a_lm, comm_map_mod.f90:228-261, and its mpi_dot_product,
 * comm_utils.f90:599-614): inside cmdr_solve rank `rank` of `nranks` keeps only its contiguous range of x, r, d, q, s.  The
 * matvec output is reduce-scattered instead of all-reduced (ncclReduceScatter with the native communicator), the three
 * fused vector kernels run on the slice, S^1/2 d is all-gathered for the next synthesis (ncclAllGather) and every dot
 * product sums its 8 KB of block partials over the ranks.  Same wire volume as the all-reduce, 1/nranks of the vector work.
 * Applies to one diffuse component with one map under the diagonal preconditioner (S^1/2 and M^-1 are then scalars per l,
 * any index range is a valid slice); other systems keep replicated vectors.  With the all-reduce callbacks the two
 * collectives are formed from all-reduces (correct, no traffic saving).  nranks <= 1 switches it off. */
int cmdr_ctx_set_vector_slicing(cmdr_ctx* ctx, int rank, int nranks);
int cmdr_ctx_rccl_size(cmdr_ctx* ctx);
int cmdr_ctx_set_only_pol(cmdr_ctx* ctx, int only_pol);
/* on != 0: reproduce cr_matmulA's re-use of pmap%alm across the components of a band literally
 * (comm_cr_mod.f90:846-861 with set_alm, comm_map_mod.f90:1193-1210): a component with a smaller lmax_amp than an
 * earlier active one of compList then feeds getBand with the EARLIER component's coefficients above its own lmax, and A
 * is not symmetric.  Needed to match a reference run with mixed lmax_amp bit for bit; default 0 = the intended
 * zero-filled semantics (identical whenever all active components share one lmax_amp).  Constant-mixing components. */
int cmdr_ctx_set_literal_quirks(cmdr_ctx* ctx, int on);

/* data(i): comm_data_mod.f90:33-63.  siN = 1/rms (0 in masked pixels, comm_N_rms_mod.f90:179-193),
 * [npix_local x nmaps]; b_l(0:lmax, nmaps) (comm_B_bl_mod.f90); sg_mask = samp_group_mask or NULL
 * (comm_N_rms_mod.f90:264-313); wring[2*nside] or NULL.  Returns the 0-based band index (< 0 on error). */
int cmdr_band_add(cmdr_ctx* ctx, int nside, int lmax, int nmaps, const double* siN, const double* b_l,
                  double mb_eff, const double* sg_mask, const double* wring);
/* One comm_diffuse_comp: lmax_amp, nmaps; lmax_cl < 0 means cltype == 'none'; sqrtS_mat / sqrtInvS_mat / S_mat are
 * (nmaps, nmaps, 0:lmax_cl) as comm_Cl%updateS leaves them (comm_Cl_mod.f90:316-384); F_mean(numband, nmaps) is
 * c%F_mean(:, 0, :) (comm_diffuse_comp_mod.f90:1991-1999); active = c%active_samp_group(samp_group).
 * Returns the 0-based component index (< 0 on error). */
int cmdr_comp_add(cmdr_ctx* ctx, int lmax_amp, int nmaps, int lmax_cl, const double* sqrtS_mat,
                  const double* sqrtInvS_mat, const double* S_mat, const double* F_mean, int active);
int cmdr_finalize(cmdr_ctx* ctx);
/* Dense QU noise covariance of a T,Q,U band (comm_N_QUcov: low-resolution WMAP-type polarisation bands): iN and its
 * symmetric square root siN_mat, (2 npix)^2 doubles each on the stacked (Q; U) pixels; N^-1 and N^-1/2 then zero the
 * temperature and multiply (Q; U) by the matrices (comm_N_QUcov_mod.f90:320-385).  The siN passed to cmdr_band_add
 * plays siN_diag (:256-272) and only feeds the preconditioner.  Before cmdr_finalize; such bands are not ring-sharded. */
int cmdr_band_set_qucov(cmdr_ctx* ctx, int band, const double* iN, const double* siN_mat);
/* Compact components in the solve (templates: comm_template_comp_mod.f90:210-270; point sources:
 * comm_ptsrc_comp_mod.f90:336-428): a block of nparam scalar amplitudes with Gaussian prior (mean, sigma) = P_cg / P_x,
 * i.e. S^1/2 = sigma (comm_cr_mod.f90:817-833).  Call cmdr_compact_add in compList order relative to cmdr_comp_add:
 * that order is the stacked-vector order.  cmdr_compact_set_band gives, for one band, what evalTemplateBand /
 * evalPtsrcBand add to the band's map per unit amplitude, as COO triplets (cell = pix_local + npix_local * stokes,
 * param, value); projectTemplateBand / projectPtsrcBand are its transpose.  Both before cmdr_finalize.  The
 * preconditioner block of a compact block is its exact dense sub-matrix of A (diagonal preconditioner type only). */
int cmdr_compact_add(cmdr_ctx* ctx, int nparam, const double* sigma, const double* mean, int active);
int cmdr_compact_set_band(cmdr_ctx* ctx, int block, int band, int64_t nnz, const int64_t* cell, const int* param,
                          const double* val);
/* Spatially varying mixing matrix F(band,0)%p%map of one component at one band (producer: updateDiffuseMixmat,
 * comm_diffuse_comp_mod.f90:1662-2023; stays on the Fortran side): npix_local x nmaps, nmaps = min(component, band).
 * The pair then takes the Y . F . YtW branch of evalDiffuseBand / projectDiffuseBand / cr_computeRHS
 * (:2082-2084, :2155-2157, comm_cr_mod.f90:640-650) instead of the F_mean fast path.  F = NULL returns to F_mean.
 * May be called before or after cmdr_finalize, and again after every mixing-matrix update. */
int cmdr_comp_set_mixing_map(cmdr_ctx* ctx, int comp, int band, const double* F, int nmaps);
/* Optional comm_Cl%getCl(l, p) table [(lmax_cl+1) x nmaps] used by the pseudo-inverse preconditioner
 * (comm_Cl_mod.f90:1440-1456: D_l 2pi/(l(l+1)) without the RJ2unit factors); default: S_mat(p,p,l). */
int cmdr_comp_set_cl_diag(cmdr_ctx* ctx, int comp, const double* cl);
int64_t cmdr_ncr(const cmdr_ctx* ctx);                    /* comm_cr_utils.f90:28 ncr */
int64_t cmdr_band_npix(const cmdr_ctx* ctx, int band);    /* local pixels per Stokes column */

/* initDiffPrecond_diagonal (comm_diffuse_comp_mod.f90:1167-1252) incl. compute_invN_lm (comm_N_mod.f90:127-197),
 * then updateDiffPrecond_diagonal (:1313-1557). */
int cmdr_precond_init_diag(cmdr_ctx* ctx);
int cmdr_precond_update_diag(cmdr_ctx* ctx);
/* Low-l dense preconditioner of one diffuse component (CG_LMAX_PRECOND = lmax_pre_lowl >= 0 with the diagonal type;
 * the reference enables it for the CMB component: comm_diffuse_comp_mod.f90:217-225).  updateLowlPrecond (:5098-5251)
 * probes the (lmax_pre_lowl+1)^2 temperature block of A with unit vectors through low-resolution transforms
 * (nside = N%nside_chisq_lowres, lmax = 2 lmax_pre_lowl) and the coadded noise InvN_lowres, adds the unit prior term
 * and Cholesky-inverts; applyLowlPrecond (:5254-5310, called from cr_invM, comm_cr_mod.f90:1058-1073) overwrites the
 * l <= lmax_pre_lowl temperature entries of M^-1 x with that inverse applied to the same entries of x.
 *   nside_lowres[band], siN_lowres[band]: data(band)%N%siN_lowres (comm_N_rms_mod.f90:250-259: sqrt(udgrade(siN^2)) *
 *   nside/nside_lowres), full-sky RING temperature map of 12 nside_lowres^2 doubles, produced by the driver (HEALPix
 *   udgrade stays on the Fortran side like every other per-band data product).
 * After cmdr_finalize; the block is (re)built by every cmdr_precond_update_diag, as update_precond does
 * (comm_cr_mod.f90:1136-1147).  lmax_pre_lowl < 0 switches it off again. */
int cmdr_precond_set_lowl(cmdr_ctx* ctx, int comp, int lmax_pre_lowl, const int* nside_lowres,
                          const double* const* siN_lowres);
/* Pseudo-inverse preconditioner, cg_precond = 'pseudoinv': alpha_nu of every band (comm_N_rms_mod.f90:217-246),
 * then updateDiffPrecond_pseudoinv (comm_diffuse_comp_mod.f90:1560-1658; SVD pseudo-inverse math_tools.f90:234-292).
 * cr_invM then runs applyDiffPrecond_pseudoinv (:2238-2380).  Whichever of the two update calls ran last selects the
 * preconditioner type used by cmdr_invM / cmdr_solve. */
int cmdr_precond_init_pseudoinv(cmdr_ctx* ctx);
int cmdr_precond_update_pseudoinv(cmdr_ctx* ctx);
int cmdr_get_alpha_nu(cmdr_ctx* ctx, int band, double* out_host /* [nmaps] */);
/* copy out data(band)%N%invN_diag%alm (nalm x nmaps) -- for parity tests */
int cmdr_get_invN_diag(cmdr_ctx* ctx, int band, double* out_host);

int cmdr_matmulA(cmdr_ctx* ctx, const double* x, double* y);            /* cr_matmulA  comm_cr_mod.f90:771-1024 */
int cmdr_invM(cmdr_ctx* ctx, const double* x, double* y);               /* cr_invM     comm_cr_mod.f90:1026-1077 */
int cmdr_matmulA_dev(cmdr_ctx* ctx, const double* x_dev, double* y_dev);
int cmdr_invM_dev(cmdr_ctx* ctx, const double* x_dev, double* y_dev);
/* cr_computeRHS (comm_cr_mod.f90:542-769).  sample != 0: operation == 'sample'.
 *   resid[i] : compute_residual(i) map of band i (comm_chisq_mod.f90:196-267), [npix_local x nmaps]
 *   xi[i]    : the unit Gaussians the Fortran side drew for band i in its own order (:602-608); ignored if !sample
 *   eta      : (ncr) unit Gaussians of the prior term in stacked order (:704-709), caller zeroes what only_pol skips
 *   mu       : (ncr) prior mean amplitudes c%mu in stacked order, or NULL (:712-725) */
int cmdr_compute_rhs(cmdr_ctx* ctx, int sample, const double* const* resid, const double* const* xi,
                     const double* eta, const double* mu, double* rhs);
int cmdr_compute_rhs_dev(cmdr_ctx* ctx, int sample, const double* const* resid_dev, const double* const* xi_dev,
                         const double* eta_dev, const double* mu_dev, double* rhs_dev);
/* compute_residual(band, cg_samp_group) for every band (comm_chisq_mod.f90:196-267, called at comm_cr_mod.f90:568):
 *   resid[i] = data[i] - signal of the components whose active flag is NOT set (those outside the sampling group),
 * i.e. Y of the summed c%getBand(alm_out) of the diffuse ones plus the pixel-space getBand of templates / point sources.
 * amp = the components' own amplitudes c%x in the stacked layout of cr_amp2x, physical units (before S^-1/2);
 * data[i], resid[i]: [npix_local x nmaps] like the maps of cmdr_compute_rhs. */
int cmdr_compute_residual(cmdr_ctx* ctx, const double* amp, const double* const* data, double* const* resid);
int cmdr_compute_residual_dev(cmdr_ctx* ctx, const double* amp_dev, const double* const* data_dev,
                              double* const* resid_dev);
/* applyMonoDipolePrior (comm_diffuse_comp_mod.f90:5738-5827), the tail of sample_amps_by_CG after cr_x2amp
 * (comm_signal_mod.f90:186-194) for a diffuse component with mono_prior_type /= 'none'
 * (COMP_MONOPOLE_PRIOR = '<type>:<mask file>', comm_diffuse_comp_mod.f90:352-356): beam-convolve the just-solved
 * temperature a_lm with the output beam, synthesise the map (map%Y), fit a monopole as the mask-weighted mean
 * (type 1 'monopole', :5761-5768) or monopole + dipole by least squares over the pixels with mask >= 0.5 (type 2
 * 'monopole+dipole', :5775-5794) and subtract the fit from the (0,0), (1,-1), (1,0), (1,1) coefficients (:5811-5824).
 *   comp     : diffuse component index as returned by cmdr_comp_add
 *   amp      : the stacked amplitudes (ncr) as cmdr_solve left them (physical units); edited in place
 *   nside    : the component's own map resolution (x%info%nside); rank-local rings as named by cmdr_ctx_set_rings
 *   b_l_out  : host, B_out%b_l(0:lmax_amp, 1) * mb_eff, or NULL for no output beam
 *   mask     : mono_prior_map%map(:, 1) on the local pixels of `nside` (npix_local doubles)
 *   mu[4]    : out, host: the fit (monopole, dipole x, y, z) the reference prints (:5773, :5802)
 * With several ranks the sums are reduced through the context's communicator / all-reduce callback, like the
 * mpi_allreduce at :5766-5767, :5792-5793.  'crosscorr' stops the reference too (:5804-5806).  The reference also
 * subtracts the fit from self%x%map, a pixel buffer of the driver the CR path never reads: not mirrored. */
int cmdr_apply_mono_dipole_prior(cmdr_ctx* ctx, int comp, double* amp, int nside, const double* b_l_out,
                                 const double* mask, int64_t npix_local, int type, double* mu);
int cmdr_apply_mono_dipole_prior_dev(cmdr_ctx* ctx, int comp, double* amp_dev, int nside, const double* b_l_out,
                                     const double* mask_dev, int type, double* mu);
/* solve_cr_eqn_by_CG (comm_cr_mod.f90:48-406).  crit: 0 'residual', 1 'fixed_iter', 2 'chisq' (cpar%cg_conv_crit;
 * 'chisq' = relative change of cr_compute_chisq, :223-226, :239-242, :408-465, evaluated against the residual maps the
 * last cmdr_compute_rhs call received -- the _dev form copies them when such a solve is entered, so they only have to be
 * alive until then -- with tol as the limit);
 * tol = cg_tol, miniter = cg_miniter, maxiter = cg_samp_group_maxiter, check_freq = cg_check_conv_freq;
 * x0 = NULL <=> cg_init_zero, else the current amplitudes (cr_amp2x).  On return x is already multiplied by
 * sqrt(S) (:350-389).  stat follows :392-395 (1 = not converged within maxiter; the caller may ignore it, as
 * comm_signal_mod.f90:181 does).  res[0] = final r^T M^-1 r, res[1] = delta0 = b^T M^-1 b. */
int cmdr_solve(cmdr_ctx* ctx, const double* b, double* x, int crit, double tol, int miniter, int maxiter,
               int check_freq, const double* x0, int* niter, double* res, int* stat);
int cmdr_solve_dev(cmdr_ctx* ctx, const double* b_dev, double* x_dev, int crit, double tol, int miniter,
                   int maxiter, int check_freq, const double* x0_dev, int* niter, double* res, int* stat);

/* Updates between Gibbs iterations (all after cmdr_finalize; follow with cmdr_precond_update_*):
 *  - cmdr_comp_set_cl: new sqrtS_mat / sqrtInvS_mat / S_mat of a component after sampleCls -> updateS
 *    (comm_Cl_mod.f90:838-863, 316-384); same shapes as at cmdr_comp_add.
 *  - cmdr_comp_set_f_mean: new F_mean(numband, nmaps) after a spectral-index update (comm_diffuse_comp_mod.f90:1991-1999);
 *    run cmdr_precond_init_* again afterwards, as the reference does (recompute_diffuse_precond).
 *  - cmdr_comp_set_active / cmdr_compact_set_active: c%active_samp_group(samp_group) when the driver moves to the
 *    next sampling group (comm_signal_mod.f90 sample_amps_by_CG). */
int cmdr_comp_set_cl(cmdr_ctx* ctx, int comp, const double* sqrtS_mat, const double* sqrtInvS_mat, const double* S_mat);
int cmdr_comp_set_f_mean(cmdr_ctx* ctx, int comp, const double* F_mean);
int cmdr_comp_set_active(cmdr_ctx* ctx, int comp, int active);
int cmdr_compact_set_active(cmdr_ctx* ctx, int block, int active);

/* getSigmaL (commander3/src/comm_map_mod.f90:1302-1351): sigma_l(0:lmax, nspec), nspec = nmaps(nmaps+1)/2 in
 * Commander's (1,1),(1,2)..(nmaps,nmaps) order, from a packed a_lm(0:nalm-1, nmaps) -- the power-spectrum statistic the
 * C_l Gibbs step (sample_powspec, commander.f90:229) consumes right after the amplitude solve. */
int cmdr_sigma_l(const double* alm, int lmax, int nmaps, double* sigma_l);
int cmdr_sigma_l_dev(const double* alm_dev, int64_t stride, int lmax, int nmaps, double* sigma_l_dev);

/* The rest of the C_l Gibbs step (commander.f90:229 sample_powspec), host side -- O(lmax) scalar work on matrices of
 * order <= 3 that the reference runs on rank 0 only; no GPU needed.
 *  - cmdr_cl_update_S: comm_Cl%updateS (comm_Cl_mod.f90:316-384).  Dl(0:lmax, nspec) and RJ2unit(nmaps) in,
 *    sqrtS_mat / sqrtInvS_mat / S_mat (nmaps, nmaps, 0:lmax) out.  Returns the number of multipoles whose matrix was not
 *    positive definite (compute_hermitian_root then leaves A(1,1) = -1e30, math_tools.f90:640-648), < 0 on error.
 *  - cmdr_cl_sample_binned: sample_Cls_inverse_wishart2 for cltype 'binned' without the lookup branch
 *    (comm_Cl_mod.f90:1008-1249 with sample_InvSamp, InvSamp_mod.f90:35-294).  bins = the bins2 tree flattened in the
 *    depth-first order sample_Dl_bin visits it; sample != 0 where stat == 'S'; spec is 1-based (TT, TE, TB, EE, EB, BB).
 *    sigma_l as cmdr_sigma_l returns it, S_mat as the last updateS left it; Dl is updated in place.  The reference draws
 *    one rand_uni per sampled bin (InvSamp_mod.f90:258-261): pass them in `uniform` in that order (the Fortran driver
 *    keeps its planck_rng handle); *nused returns how many were consumed.  Returns 0, 1 for the reference's
 *    ok = .false. (a bin's sampler failed; later bins untouched), < 0 on error. */
typedef struct cmdr_cl_bin {
    int lmin, lmax, spec, sample;
    double sigma;
} cmdr_cl_bin;
int cmdr_cl_update_S(int lmax, int nmaps, int lmin, const double* Dl, const double* RJ2unit, double* sqrtS_mat,
                     double* sqrtInvS_mat, double* S_mat);
int cmdr_cl_sample_binned(int lmax, int nmaps, const double* sigma_l, const double* S_mat, const double* RJ2unit, int nbin,
                          const cmdr_cl_bin* bins, const double* uniform, int nuniform, double* Dl, int* nused);
/* sample_Dl_lookup, the table branch of the binned sampler (comm_Cl_mod.f90:1063-1145; runs before the bins when
 * lmin_lookup >= 0): nmodel tabulated spectra Dl_lookup(lmin_lookup:lmax_lookup, 6, nmodel), active(6) = which of
 * TT,TE,TB,EE,EB,BB they replace; one of them is drawn with probability proportional to its inverse-Wishart likelihood
 * given sigma_l, using ONE uniform variate, and copied into Dl.  Polarised components (nmaps = 3) only, like the
 * reference.  *chosen = 0-based model index.  Returns 0, 1 (every model failed: ok = .false.), < 0 on error. */
int cmdr_cl_sample_lookup(int lmax, int lmin_lookup, int lmax_lookup, int nmodel, const double* Dl_lookup,
                          const int* active, const double* sigma_l, const double* S_mat, const double* RJ2unit,
                          double uniform, double* Dl, int* chosen);
/* get_Cl_apod (comm_Cl_mod.f90:676-704): the per-l factor matmulSqrtS / matmulS / matmulSqrtInvS / getCl apply on top of
 * the updateS tables (:572-666, :1454) -- not 1 for l < COMP_PRIOR_AMP_LMAX (cs_lmax_amp_prior, :134).  The solver context
 * takes tables with the factor folded in: cmdr_cl_apply_apod scales sqrtS_mat by f, S_mat by f^2 and sqrtInvS_mat by 1/f
 * (0 where f = 0) in place; call it on a COPY of the updateS output before cmdr_comp_add / cmdr_comp_set_cl
 * (cmdr_cl_sample_binned wants the raw S_mat, as the reference's lnL_invWishart does). */
double cmdr_cl_apod(int l, int l_apod, int lmax, int lmax_prior, int positive);
int cmdr_cl_apply_apod(int lmax, int nmaps, int l_apod, int lmax_prior, double* sqrtS_mat, double* sqrtInvS_mat,
                       double* S_mat);

/* HIP-event timing of the dominant kernels, on the stream they are launched on.  kinds: 0 Legendre synthesis
 * launches, 1 fused ring-stage launches, 2 Legendre adjoint launches, 3 whole cr_matmulA.  ms_sum[4], count[4]. */
int cmdr_profile_enable(cmdr_ctx* ctx, int on);
int cmdr_profile_read(cmdr_ctx* ctx, double* ms_sum, long long* count);
/* the same with nkinds <= 8 entries: kind 4 = launches of the matrix-unit Legendre adjoint kernel (k_leg_adj_mx, up to 8
 * maps per launch: the dominant kernel), kind 5 = the VALU adjoint launches of the maps it leaves over; 4 + 5 = the
 * scalar part of kind 2; kinds 6 / 7 = the spin-2 (Q,U) synthesis / adjoint launches of a polarised plan (parts of 0 / 2) */
int cmdr_profile_read_ext(cmdr_ctx* ctx, int nkinds, double* ms_sum, long long* count);
/* out[0] = number of (band, Stokes) maps, out[1] = wave tasks of the first plan, out[2] = total (ring pair, l, m)
 * recursion steps one Legendre launch of the first plan performs for ONE map (algorithmic work, mlim-pruned) */
int cmdr_problem_info(cmdr_ctx* ctx, int64_t* out);
/* n <= 8 entries: [0..2] as above, [3] = the same step count for ONE (Q,U) pair of the first plan's spin-2 launches (0:
 * unpolarised), [4] / [5] = scalar maps / polarisation pairs of the first plan, [6] = its local ring pairs, [7] = plans */
int cmdr_problem_info_ext(cmdr_ctx* ctx, int n, int64_t* out);

/* Chain-file order of component amplitudes (the "alm" dataset written by comm_map%writeFITS into the HDF chain file,
 * comm_map_mod.f90:712-740: single precision, index l^2 + l + m with m = -l..l, nmaps columns) <-> Commander's packed
 * a_lm columns; lets chains produced through this library be diffed against reference chains and be restarted from
 * them.  Host pointers; alm: (lmax+1)^2 x nmaps doubles, chain32: (lmax+1)^2 x nmaps floats. */
int cmdr_alm_to_chain_order(const double* alm, int lmax, int nmaps, float* chain32);
int cmdr_alm_from_chain_order(const float* chain32, int lmax, int nmaps, double* alm);

/* Chain-file I/O of one component's amplitude sample, in the layout Commander's HDF5 chain files
 * have, so that chains produced through this library can be diffed against reference chains and restarted from either:
 *   /<iter, 6 digits>/<label>/amp_alm (float32, index l^2 + l + m, nmaps columns), amp_lmax, amp_nmaps (int32),
 *   sigma_l and Dl (float64, (0:lmax, nspec))
 * -- comm_diffuse_comp%dumpFITS / comm_map%writeFITS (comm_diffuse_comp_mod.f90:2459-2524, comm_map_mod.f90:712-745),
 * write_Dl_to_FITS (comm_Cl_mod.f90:1335); read back by initDiffuseHDF / readHDF / comm_Cl%initHDF
 * (comm_diffuse_comp_mod.f90:2687-2728, comm_map_mod.f90:860-889, comm_Cl_mod.f90:1393).
 * alm: packed a_lm (nalm x nmaps) in the units of c%x; unit_scale[nmaps] = RJ2unit_ * cg_scale (written values are
 * alm * unit_scale, read values are divided by it; NULL = 1).  sigma_l / Dl: (lmax+1) x nspec column-major or NULL.
 * The file is created if it does not exist; an existing group / dataset of the same iteration is replaced.
 * HDF5 (>= 1.10) is loaded at run time (dlopen libhdf5; CMDR_HDF5_LIB overrides the search); host pointers. */
int cmdr_chain_write_comp(const char* chainfile, int iter, const char* label, const double* alm, int lmax, int nmaps,
                          const double* unit_scale, const double* sigma_l, const double* Dl);
int cmdr_chain_read_comp(const char* chainfile, int iter, const char* label, int lmax, int nmaps,
                         const double* unit_scale, double* alm, double* Dl);

#ifdef __cplusplus
}
#endif
#endif
