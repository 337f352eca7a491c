// The constrained-realization linear system on the device: what commander3/src/comm_cr_mod.f90 does per Gibbs
// amplitude sample, with every vector resident in HBM (persistent workspaces; the reference allocates and frees
// its temporaries on every call, comm_cr_mod.f90:847-848,952-953).
//
//   A = 1 + S^1/2 [ sum_bands F^T B^T Y^T N^-1 Y B F ] S^1/2        cr_matmulA        comm_cr_mod.f90:771-1024
//   b = S^1/2 sum_bands F^T B^T Y^T (N^-1 d | N^-1/2 (N^-1/2 d + xi)) + eta (+ S^-1/2 mu)   cr_computeRHS :542-769
//   M^-1 = block-diagonal in (l, m, stokes) over components        cr_invM :1026-1077, diagonal type
//   PCG (Shewchuk form)                                            solve_cr_eqn_by_CG :48-406
//
// Scope: diffuse components with constant mixing (F_mean fast path, comm_diffuse_comp_mod.f90:2077-2080) or spatially
// varying mixing (Y . F . YtW branch, :2082-2084, :2155-2157), white per-pixel noise (comm_N_rms), T and T,Q,U bands;
// diagonal and pseudo-inverse preconditioners.
#pragma once
#include <map>
#include <memory>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"
#include "rccl_dyn.hpp"
#include "sht_plan.hpp"

namespace cmdr {

// In-place sum over ranks of a device vector (ring / band sharding across GPUs).  Called on the host thread with
// the library stream idle; must return with the result complete.  The host language supplies it
// (torch.distributed / RCCL in bench.py, MPI in the Fortran driver).
using AllreduceFn = void (*)(void* user, double* dev_ptr, int64_t n);
// Stream-ordered variant: enqueue the sum on the library stream and return (RCCL); no host synchronisation.
using AllreduceStreamFn = void (*)(void* user, double* dev_ptr, int64_t n, void* hip_stream);

struct SolveResult {
    int niter = 0;
    int stat = 0;
    double delta_new = 0.0, delta0 = 0.0;
};

class CrSystem {
  public:
    explicit CrSystem(int device);
    ~CrSystem();

    void set_rings(int nside, const std::vector<int>& rings);  // multi-GPU ring subset for bands of this nside
    int add_band(int nside, int lmax, int nmaps, const double* siN, const double* b_l, double mb_eff,
                 const double* sg_mask, const double* wring);
    int add_comp(int lmax_amp, int nmaps, int lmax_cl, const double* sqrtS, const double* sqrtInvS,
                 const double* S, const double* F_mean, int active);
    // Compact components (templates, point sources): nparam scalar amplitudes with Gaussian prior (mean, sigma); call in
    // compList order relative to add_comp (that order is the stacked-vector order).  Then one sparse matrix per band:
    // COO triplets (cell = pix_local + npix_local * stokes, param, value) = what evalTemplateBand / evalPtsrcBand add
    // to the band's map per unit amplitude.
    // comm_N_QUcov (comm_N_QUcov_mod.f90): dense inverse noise covariance and its symmetric square root on the stacked
    // (Q; U) pixels of a T,Q,U band (temperature is zeroed); the band's siN then plays siN_diag (preconditioner only).
    void set_band_qucov(int band, const double* iN, const double* siN_mat);
    int add_compact(int nparam, const double* sigma, const double* mean, int active);
    void set_compact_band(int block, int band, int64_t nnz, const int64_t* cell, const int* param, const double* val);
    void finalize();
    // F(band,0)%p%map of a component with spatially varying mixing: npix_local x nmaps (nmaps = min of the band's and
    // the component's), or nullptr to go back to the F_mean fast path.  Callable before or after finalize.
    void set_mixing_map(int comp, int band, const double* F, int nmaps);
    void set_comp_cl(int comp, const double* sqrtS, const double* sqrtInvS, const double* S);
    void set_comp_f_mean(int comp, const double* F_mean);
    void set_active(int kind, int idx, int active);
    void compute_residual(const double* amp, const double* const* data, double* const* resid);   // device pointers
    // applyMonoDipolePrior (comm_diffuse_comp_mod.f90:5738-5827; called from sample_amps_by_CG,
    // comm_signal_mod.f90:186-194) on the amplitudes a chain keeps on the device: amp = the stacked vector after
    // cr_x2amp (device), mask = the component's mono_prior_map, temperature column, local pixels of the rings
    // cmdr_ctx_set_rings named for `nside` (device); b_l_out = the output beam B_out%b_l(0:lmax, 1) * mb_eff (host,
    // nullptr = 1).  type 1 'monopole', 2 'monopole+dipole'.  mu[4] = the fitted (monopole, x, y, z dipole), as printed
    // by the reference.  Edits the (0,0), (1,-1), (1,0), (1,1) temperature entries of the component in place.
    void apply_mono_dipole_prior(int comp, double* amp, int nside, const double* b_l_out, const double* mask, int type,
                                 double* mu);
    void set_cl_diag(int comp, const double* cl);   // getCl(l, p): (lmax_cl+1) x nmaps, for the pseudo-inverse U
    void set_allreduce(AllreduceFn fn, void* user) { allreduce_ = fn; allreduce_user_ = user; }
    void set_allreduce_stream(AllreduceStreamFn fn, void* user) { allreduce_s_ = fn; allreduce_s_user_ = user; }
    // Band x ring-set hybrid sharding: this rank holds a SUBSET of the bands (all components), on a subset of the
    // rings.  `rings_fn` sums over the ranks that share this rank's bands (its ring group): per-band quantities
    // (noise a_lm, mixing intermediates); the main callback sums over all ranks (the stacked vector, which is a sum
    // over bands and rings).  ring_replicas = ranks per ring group (the band-summed preconditioner matrix is then
    // counted that many times by the world sum).
    void set_band_sharding(AllreduceFn rings_fn, void* user, int ring_replicas) {
        allreduce_rings_ = rings_fn; allreduce_rings_user_ = user; ring_replicas_ = ring_replicas; band_sharded_ = true;
    }
    void set_only_pol(bool v) { only_pol_ = v; }
    // Reproduce cr_matmulA's buffer re-use literally (comm_cr_mod.f90:846-861): pmap%alm is allocated once per band and
    // set_alm (comm_map_mod.f90:1193-1210) only overwrites the (l, m) a component HAS, so a component with a smaller
    // lmax_amp than an earlier one of the list feeds getBand with the earlier component's coefficients above its own
    // lmax.  Default off: the intended zero-filled semantics (deviation 1).  Constant-mixing components only.
    void set_literal_quirks(bool v);
    // RCCL inside the library: every sum over ranks becomes an ncclAllReduce enqueued on the library stream (no
    // callback, no host synchronisation).  id = the 128-byte ncclUniqueId one rank created and the host language
    // broadcast.  Takes precedence over the callbacks.  split_rings: sub-communicator of the ranks that hold the
    // same bands (band x ring-set hybrid), ncclCommSplit with color = band group; implies band sharding.
    void init_rccl(const char* id, int rank, int nranks);
    void rccl_split_rings(int band_group, int ring_index, int ring_replicas);
    int rccl_size() const { return rccl_.ready() ? rccl_.size() : 0; }
    void drop_rccl();
    // m-sliced CG vectors (ownership comm_map_mod.f90:228-261, mpi_dot_product comm_utils.f90:599-614):
    // inside a fused fixed/residual PCG loop rank `rank` of `nranks` keeps only the index range
    // [rank c, (rank + 1) c), c = ceil(ncr / nranks), of x, r, d, q, s: the matvec output is reduce-scattered instead of
    // all-reduced, the three vector kernels run on the slice, S^1/2 d is all-gathered for the next synthesis, and each dot
    // product sums its block partials over the ranks (8 KB).  One diffuse component with one map under the diagonal
    // preconditioner (the (re, im) slots are then independent); anything else keeps replicated vectors.  nranks <= 1: off.
    void set_vector_slicing(int rank, int nranks);

    int64_t ncr() const { return ncr_; }
    int nband() const { return (int)bands_.size(); }
    int64_t band_npix(int b) const;
    int64_t band_nalm(int b) const { return nalm_packed(bands_[b].lmax) * bands_[b].nmaps; }
    int band_nmaps(int b) const { return bands_[b].nmaps; }

    void precond_init_diag();     // initDiffPrecond_diagonal + compute_invN_lm
    void precond_update_diag();   // updateDiffPrecond_diagonal
    // Low-l dense preconditioner of one diffuse component (CG_LMAX_PRECOND >= 0: updateLowlPrecond / applyLowlPrecond,
    // comm_diffuse_comp_mod.f90:5098-5310, applied at comm_cr_mod.f90:1058-1073).  siN_lowres[b]: the band's
    // N%siN_lowres (comm_N_rms_mod.f90:250-259), full-sky RING at nside_lowres[b], temperature column, host pointer.
    // lmax_pre_lowl < 0 removes it.  The block is (re)built by precond_update_diag, as update_precond does.
    void set_lowl(int comp, int lmax_pre_lowl, const int* nside_lowres, const double* const* siN_lowres);
    void precond_init_pseudoinv();    // alpha_nu (comm_N_rms_mod.f90:217-246) and the N maps of the T operator
    void precond_update_pseudoinv();  // updateDiffPrecond_pseudoinv (comm_diffuse_comp_mod.f90:1560-1658)
    const std::vector<double>& alpha_nu(int band) const { return bands_[band].alpha_nu; }
    const double* invN_diag_dev(int band) const { return bands_[band].invN_diag.get(); }

    // all pointers below are device pointers
    void matmulA(const double* x, double* y);
    void invM(const double* x, double* y);
    void compute_rhs(bool sample, const double* const* resid, const double* const* xi, const double* eta,
                     const double* mu, double* rhs);
    SolveResult solve(const double* b, double* x, int crit, double tol, int miniter, int maxiter, int check_freq,
                      const double* x0);
    void sync();
    hipStream_t stream() const { return stream_; }
    // HIP-event timing of the dominant kernels on the library stream (bench.py roofline leg).
    // kinds: 0 Legendre synthesis, 1 fused ring stage, 2 Legendre adjoint, 3 whole matvec
    void set_profile(bool on);
    void read_profile(double* ms_sum, long long* count, int nkinds = 4);   // [nkinds <= 8] each; drains pending events
    void problem_info(int64_t* out) const;
    void problem_info_ext(int n, int64_t* out) const;

  private:
    struct Band {
        int nside, lmax, nmaps, group = -1;
        double mb_eff;
        std::vector<double> b_l;            // (lmax+1) x nmaps, column-major
        DevBuf<double> siN, mul;            // npix_local x nmaps: 1/rms (* samp-group mask), and siN^2 * mask
        DevBuf<double> siN_raw;             // bare 1/rms, only kept when a samp-group mask was given
        DevBuf<double> invN_diag;           // nalm x nmaps (device)
        std::vector<double> invN_diag_h;
        std::vector<double> wring;
        bool has_wring = false;
        std::vector<double> Nmap_h;         // rms^2 (* mask), 0 where siN = 0: comm_N_rms%N (comm_N_rms_mod.f90:288-301)
        DevBuf<double> mulP;                // (W_ring 4pi/Npix)^2 * Nmap: WY . N . YtW of the pseudo-inverse precond
        std::vector<double> alpha_nu;       // [nmaps]
        DevBuf<double> qucov_iN, qucov_siN; // (2 npix)^2 each, row-major (symmetric); empty: white noise
    };
    struct Comp {
        CompDev d;
        std::vector<double> sqrtS, sqrtInvS, S;  // nmaps x nmaps x (lmax_cl+1), Fortran order
        std::vector<double> F_mean;              // nband x nmaps, column-major (F_mean(band,0,stokes))
        std::vector<std::vector<double>> F_map;  // [nband]: empty, or npix_local x nm host copy of F(band,0)%p%map
        std::vector<int> F_map_nm;
        std::vector<DevBuf<double>> mulF;        // [nband]: F * W_ring 4pi/Npix (device), built by rebuild_mixing
        std::vector<char> mulF_dirty;            // [nband]: the host map changed since mulF was uploaded
        std::vector<double> cl_diag;             // optional getCl table
    };
    struct CompactBand {                         // P_b of one block on one band, both orientations
        int band = -1;
        int64_t nrows = 0;
        std::vector<int64_t> h_cell; std::vector<int> h_param; std::vector<double> h_val;   // COO until finalize
        DevBuf<int64_t> rows, rptr, cptr, ccell;   // CSR over touched cells; CSC by parameter
        DevBuf<int> rcol;
        DevBuf<double> rval, cval;
    };
    struct Compact {
        int nparam = 0, active = 1;
        int64_t pos = 0;
        std::vector<double> sigma, mean;
        DevBuf<double> sigma_dev, mean_dev, Minv;
        std::vector<CompactBand> P;
    };
    struct LowL {
        int comp = -1, L = -1;
        std::vector<int> nside;                       // [nband]
        std::vector<std::vector<double>> iN;          // [nband] siN_lowres^2, full sky
        std::vector<std::unique_ptr<DevBuf<double>>> iN_dev;
        DevBuf<double> Minv, xl, yl;                  // (L+1)^4, (L+1)^2, (L+1)^2
        DevBuf<int64_t> idx;                          // stacked-vector position of (l, m), chain order l^2 + l + m
        bool ready = false;
    };
    std::vector<LowL> lowl_;
    std::map<int, std::unique_ptr<ShtPlan>> lowl_plans_;   // by nside_lowres * 65536 + 2 L
    std::map<int64_t, std::unique_ptr<ShtPlan>> md_plans_; // applyMonoDipolePrior at (nside, lmax) no band plan has
    DevBuf<double> md_alm_, md_map_, md_part_, md_bl_;
    void lowl_update(LowL& W);
    struct MixCol { int bm, comp, stokes; };     // one scalar column / first column of a (Q,U) pair of a mixing batch
    struct MixBatch {
        std::vector<MixCol> T, P;
        DevBuf<const double*> mul_ptrs;          // [nT + 2 nP]
    };
    struct Group {
        int nside, lmax, nbm = 0;
        int nT = 0, npol = 0;              // maps ordered [T of every band][(Q,U) of every polarised band]
        std::vector<int> bands, bm_band, bm_stokes;
        std::unique_ptr<ShtPlan> plan;
        DevBuf<double> w;                   // [nbm][ncomp][lmax+1]
        DevBuf<double> w_fwd;               // forward weights of cr_matmulA when literal_quirks_ is set (else unused)
        DevBuf<int> bm_stokes_dev;
        DevBuf<const double*> mul_ptrs;     // [nbm]
        DevBuf<cd> that;                    // [nbm][that_elems]: circulant spectra of the mul maps (Toeplitz ring form)
        DevBuf<double> tmpmap;              // [nbm][npix_local] (RHS only; allocated lazily)
        std::vector<hipEvent_t> ev_synth, ev_ring;   // pipelined matvec: per batch of maps
        bool ring_pending = false;          // the adjoint must wait for ev_ring
        DevBuf<double> bl;                  // [nbm][lmax+1]  b_l * mb_eff of each band map
        // spatially varying mixing
        std::vector<MixBatch> mix;
        DevBuf<double> mix_in, mix_out;     // [max_maps][nalm]
        DevBuf<double> E, U;                // [nbm][nalm]: mixed band signal in, Yt N^-1 Y output
        // pseudo-inverse preconditioner
        DevBuf<double> w_pin, w_pout;       // [nbm][ncomp][lmax+1]
        DevBuf<const double*> mulP_ptrs;    // [nbm]
        DevBuf<cd> thatP;                   // circulant spectra of the mulP maps (Toeplitz ring form of the N operator)
    };
    CellBase cell_base(int band) const;
    void compact_forward(Group& G, const double* sx, double* maps);     // maps += P (sigma already applied: sx)
    void compact_adjoint(Group& G, const double* maps, double* yc);      // yc[block] += P^t maps
    void compact_precond_init();
    bool group_has_compact(const Group& G) const;   // also true for dense-noise bands: the map has to exist
    void qucov_invN(Group& G, int band, double* maps);   // maps(T) = 0, maps(Q;U) = iN maps(Q;U)
    DevBuf<double> qucov_tmp_, qucov_tmp2_;
    void rebuild_weights();
    void flip_active();
    void matmulA_impl(const double* x, double* y, bool sx_ready, bool finish);
    void synth_T_of(Group& G, const double* v, const double* w, const double* extra);
    void forward_maps(Group& G, const double* sx);
    double chisq_of(const double* x);
    std::vector<const double*> last_resid_;   // residual maps of the last compute_rhs (caller's until a chisq solve copies them)
    std::vector<DevBuf<double>> resid_own_;
    bool resid_owned_ = false;
    void rebuild_mixing();
    void mix_forward(Group& G, const double* sx);
    void mix_adjoint(Group& G, bool rhs);
    void apply_pseudoinv(const double* x, double* y);
    void adjoint_groups_to_yc(bool from_maps);
    static constexpr int kPipeBatch = 3;   // maps per pipeline stage = maps per Legendre recursion at R = 4
    struct Span { hipEvent_t a, b; int kind; hipStream_t s; };
    void span_begin(int kind, hipStream_t s = nullptr);   // nullptr: the main stream
    void span_end();
    // Pipelined matvec (T-only plans with more than one batch of maps): the LDS-bound ring stage of batch j runs on a
    // second stream beside the VALU-bound Legendre synthesis of batch j+1 / adjoint of batch j-1.  OFF by default
    // (CMDR_PIPELINE=1 enables it): measured 12.6 ms against 12.2 ms serial at cfg3 -- a ring workgroup (8 waves,
    // 147 KB LDS) does not fit beside a CU full of Legendre waves (register file), so the streams only time-slice.
    bool pipelined(const Group& G) const;
    void pipeline_events(Group& G, int nbatch);
    hipStream_t stream_ring_ = nullptr;
    bool pipeline_ = false;
    std::vector<Span> spans_;
    std::vector<int> open_;
    // 0..3 as cmdr_profile_read; 4 matrix-unit adjoint launch, 5 VALU adjoint launches; 6 / 7 the spin-2 synthesis / adjoint
    // launches of a polarised plan (inside kinds 0 / 2)
    static constexpr int kProfKinds = 8;
    double prof_ms_[kProfKinds] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_n_[kProfKinds] = {0, 0, 0, 0, 0, 0, 0, 0};
    void reduce(double* v, int64_t n);         // over all ranks
    void reduce_rows(double* v, int m0, int m1, hipStream_t st);   // rows m0 <= m < m1 of every diffuse block, on stream st
    int slice_rank_ = 0, slice_n_ = 0;         // set_vector_slicing
    bool slice_active_ = false;                // inside a sliced PCG loop: adjoint_groups_to_yc reduce-scatters yc_
    int64_t slice_count() const { return (ncr_ + slice_n_ - 1) / slice_n_; }
    void slice_reduce_scatter(double* v);      // padded buffers of slice_n_ * slice_count() doubles, in place
    void slice_all_gather(double* v);
    hipStream_t stream_comm_ = nullptr;        // the first half's all-reduce runs here beside the second half's adjoint
    hipEvent_t ev_half_ = nullptr, ev_comm_ = nullptr;
    void reduce_rings(double* v, int64_t n);   // over the ranks holding the same bands (== reduce without band sharding)

    hipStream_t stream_ = nullptr;
    std::vector<Band> bands_;
    std::vector<Comp> comps_;
    std::vector<Compact> compacts_;
    DevBuf<double> compact_scratch_;           // slice sums of k_compact_adj
    std::vector<std::pair<int, int>> order_;   // stacked-vector order: (0, diffuse index) | (1, compact index)
    std::vector<Group> groups_;
    std::vector<std::pair<int, std::vector<int>>> ring_sets_;
    bool finalized_ = false, only_pol_ = false, profile_ = false, literal_quirks_ = false;
    int64_t ncr_ = 0;
    int lmax_max_ = -1;
    DevBuf<CompDev> comps_dev_;
    DevBuf<double> smat_;
    DevBuf<double> sx_, yc_, r_, d_, q_, s_, tmp_;
    DevBuf<double> dot_partial_, scal_;
    DevBuf<double> cg_partials_;        // fused PCG kernels: [d.q | r.s even | r.s odd] x dot_partial_count()
    // diagonal preconditioner
    int lmax_pre_ = -1, nmaps_pre_ = 0;
    std::vector<double> M0_;                // [nmaps_pre][npre][npre][ntri(lmax_pre)]
    DevBuf<double> P_;
    bool precond_ready_ = false;
    int precond_type_ = 0;                  // 0 diagonal, 1 pseudo-inverse
    DevBuf<double> Qprior_;                 // [nmaps_pre][npre][npre][lmax_pre+1]
    bool pinv_init_ = false;
    AllreduceFn allreduce_ = nullptr;
    void* allreduce_user_ = nullptr;
    AllreduceStreamFn allreduce_s_ = nullptr;
    void* allreduce_s_user_ = nullptr;
    AllreduceFn allreduce_rings_ = nullptr;
    void* allreduce_rings_user_ = nullptr;
    int ring_replicas_ = 1;
    bool band_sharded_ = false;
    RcclComm rccl_, rccl_rings_;
};

}  // namespace cmdr
