// Device-resident SHT plan: uploads ShtTables and owns the per-plan workspaces (coefficient streams, phases,
// adjoint partial columns).  One plan per distinct (nside, lmax, ring subset); Commander keeps the equivalent
// libsharp handles in its comm_mapinfo cache (commander3/src/comm_map_mod.f90:126-127,157-169).
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"
#include "plan_tables.hpp"

namespace cmdr {

class LegendreDev {  // device mirror of LegendreTables
  public:
    void upload(const LegendreTables& T);
    LegArgs args() const;        // adjoint kernel (R pairs per lane)
    LegArgs args_synth() const;  // synthesis kernel (Rs pairs per lane)
    int lmax = -1, npair_pad = 0, R = 1, Rs = 1, nchunk = 0, ntasks = 0, ntasks_s = 0;
    bool synth_wg = false;
    bool uniform_start = false;
    DevBuf<double> x, seedc, seedp, alpha, cnorm;
    DevBuf<int> ls, lw_chunk;
    DevBuf<WaveTask> tasks, tasks_s, tasks_split;
    int m_split = 0, nsplit_lo = 0, nsplit = 0;
    int64_t ph_elems() const { return (int64_t)(lmax + 1) * npair_pad * 4; }       // doubles per map
    int64_t tri_elems() const { return 2 * ntrip(lmax); }                          // doubles per map
};

class Legendre2Dev {  // device mirror of Legendre2Tables
  public:
    void upload(const Legendre2Tables& T);
    int lmax = -1, npair_pad = 0, R = 2, nchunk = 0, ntasks = 0;
    DevBuf<double> x, seed, alpha, beta, cnorm, abs_;
    DevBuf<int> ls, lw_chunk;
    DevBuf<WaveTask> tasks;
    int64_t tri4() const { return 4 * ntrip(lmax); }   // doubles per polarisation pair (stream / one partial chunk)
};

class ShtPlan {
  public:
    ShtPlan(int nside, int lmax, const std::vector<int>& rings, const double* wring, int max_maps, bool pol = false);
    bool pol() const { return pol_; }
    // (Q,U) <-> (E,B): Commander's spin-2 call on columns 2:3 (comm_map_mod.f90:446-449, 519-523, 549-553)
    void alm2map_spin2(const double* d_E, const double* d_B, double* d_Q, double* d_U, bool weighted, hipStream_t s);
    void map2alm_spin2(const double* d_Q, const double* d_U, double* d_E, double* d_B, bool weighted, hipStream_t s);
    double* stream2() { return st2_.get(); }               // [4 * ntrip][npol interleaved]
    double* partials2() { return part2_.get(); }           // [npol][nchunk2][4 * ntrip]
    int64_t part2_pol_stride() const { return (int64_t)leg2_.nchunk * leg2_.tri4(); }
    void synth2_from_stream(int npol, int kq0, hipStream_t s);     // stream2 -> phases of maps kq0 ..
    void adjoint2_to_partials(int npol, int kq0, hipStream_t s);
    const Legendre2Dev& leg2() const { return leg2_; }
    int nside() const { return T_.nside; }
    int lmax() const { return T_.lmax; }
    int64_t npix_local() const { return T_.ring.npix_local; }
    int64_t nalm() const { return nalm_packed(T_.lmax); }
    int max_maps() const { return max_maps_; }
    const ShtTables& tables() const { return T_; }

    // Commander entry points (comm_map_mod.f90:437-579), device pointers, column-major [n][nmaps] with the given
    // column strides.  weighted=false: Y / Yt ; weighted=true: WY / YtW.
    void alm2map(const double* d_alm, int64_t alm_stride, double* d_map, int64_t map_stride, int nmaps,
                 bool weighted, hipStream_t s);
    void map2alm(const double* d_map, int64_t map_stride, double* d_alm, int64_t alm_stride, int nmaps,
                 bool weighted, hipStream_t s);

    // y_k = Yt diag(mul_k) Y x_k with the map never leaving LDS: nT scalar columns followed by npol (E,B) pairs whose
    // (Q,U) maps take mul_{nT+2i}, mul_{nT+2i+1}.  d_in / d_out: nT + 2 npol packed a_lm columns (stride nalm());
    // d_mul: device array of nT + 2 npol device map pointers.  Ring weights (YtW / WY) are folded into mul by the
    // caller (pixel_weights()).  Needs nT + 2 npol <= max_maps.
    void sandwich(const double* d_in, double* d_out, const double* const* d_mul, int nT, int npol, hipStream_t s, bool share_in = false,
                  bool sum_out = false);
    // masked monopole / dipole sums of a local map (applyMonoDipolePrior): d_out[npair()][16] per ring pair
    int npair() const { return T_.ring.npair; }
    void md_sums(const double* d_map, const double* d_mask, int type, double* d_out, hipStream_t s) {
        launch_md_sums(rings_.get(), T_.ring.npair, T_.nside, d_map, d_mask, type, d_out, s);
    }
    // W_ring * 4 pi / Npix for every pixel of the local map (host)
    std::vector<double> pixel_weights() const;

    // Building blocks used by the fused CR matvec.
    double* stream() { return ast_.get(); }                // [max_maps][tri_elems]
    double* phases() { return ph_.get(); }                 // [max_maps][ph_elems]
    double* partials() { return part_.get(); }             // [max_maps][nchunk][tri_elems]
    int64_t part_map_stride() const { return (int64_t)leg_.nchunk * leg_.tri_elems(); }
    void synth_from_stream(int nmaps, hipStream_t s);                                // stream -> phases
    bool can_prep() const;                             // the synthesis can form its coefficients itself (workgroup form)
    void synth_from_prep(const PrepDev& prep, int nmaps, hipStream_t s);             // stacked vector -> phases
    // that (mode 2 only): multiplier spectra from toeplitz_build() for the same d_mul maps -> cap rings take the
    // Toeplitz form (kernels_body.hpp) instead of two Bluestein transforms per direction
    void rings(int mode, double* d_map, int64_t map_stride, const double* const* d_mul, bool weighted, int nmaps,
               hipStream_t s, const cd* that = nullptr);                             // see launch_ring
    // Circulant spectra of nmaps pixel multiplier maps (HOST array of device pointers, the maps rings(2, ...) will be
    // given): [nmaps][that_elems()] complex.  Setup-time (once per noise / mixing map); empty plan part -> size 0.
    int64_t that_elems() const { return T_.ring.that_elems; }
    void toeplitz_build(const std::vector<const double*>& mul_host, DevBuf<cd>& out, hipStream_t s);
    // phases -> partials; between(nmx) is called after the matrix-unit launches (nmx maps, 0 = none) and before the VALU ones
    void adjoint_to_partials(int nmaps, bool square, hipStream_t s, const std::function<void(int)>& between = nullptr);
    // the same three stages on maps k0 .. k0+n-1 of a stream holding nbs maps (pipelined matvec: the ring stage of
    // one batch runs beside the Legendre stage of the next)
    void synth_range(int k0, int n, int nbs, hipStream_t s);
    void rings_fused_range(int k0, int n, const double* const* d_mul, hipStream_t s);
    void adjoint_range(int k0, int n, hipStream_t s);
    // the scalar adjoint of the columns m < m_split() (half 0) or m >= m_split() (half 1) only
    void adjoint_half_to_partials(int nmaps, int half, hipStream_t s);
    int m_split() const { return leg_.m_split; }
    const LegendreDev& leg() const { return leg_; }

  private:
    ShtTables T_;
    LegendreDev leg_;
    Legendre2Dev leg2_;
    int max_maps_;
    bool pol_ = false;
    DevBuf<double> st2_, part2_;
    DevBuf<RingDev> rings_, rings_t2_;           // rings_t2_: Toeplitz pairs with mmax_eff doubled (t_d setup transform)
    std::vector<DevBuf<int>> cls_, cls_t_, cls_tb_, cls_ts_;
    std::vector<int> ncls_, ncls_t_, ncls_tb_, ncls_ts_;
    DevBuf<double> tw_, chirp_, ring_scratch_;   // ring_scratch_: one line of n/2 complex per split ring pair and map
    DevBuf<double> ast_, ph_, part_;
    DevBuf<double> share_map_;   // pixel maps of sandwich()'s shared scalar columns (allocated on first use)
};

}  // namespace cmdr
