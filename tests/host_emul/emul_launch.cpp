// TEST INFRASTRUCTURE ONLY: host implementations of every launch_* entry of commander_amd/csrc/kernels.hpp as
// plain loops over the shared kernel bodies (kernels_body.hpp, cr_body.hpp).  Linked with the real host code
// (plan_tables.cpp, sht_plan.cpp, cr_system.cpp, c_api.cpp) into tests/host_emul/_build/libcmdr_emul.so so the
// CPU-only test tier can check index maps, normalisations and the CR orchestration against the oracle.
// Not a fallback: libcmdr_hip.so never contains or loads this.
#include <algorithm>
#include <vector>

#include "kernels.hpp"

namespace cmdr {

int leg_max_batch(int R) { return R == 1 ? 9 : (R == 2 ? 4 : 3); }

template <int R, int NB>
static void synth_RN(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, int nbs, int k0,
                     double* ph, int64_t ph_stride) {
    for (int t = 0; t < ntasks; ++t)
        for (int lane = 0; lane < 64 && tasks[t].chunk >= 0; ++lane)
            leg_synth_lane<R, NB>(A, ast, nbs, k0, ph, ph_stride, tasks[t].m, tasks[t].chunk, tasks[t].lw,
                                  tasks[t].lAend, lane);
}
bool leg_synth_can_prep(const LegArgs& A) { return A.wg != 0; }
void launch_leg_synth(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, double* ph,
                      int64_t ph_stride, int nmaps, hipStream_t, int nbs, const PrepDev* prep) {
    if (nbs < 0) nbs = nmaps;
    std::vector<double> formed;
    if (prep) {   // what the kernel's tile staging computes, element by element
        const int lmax = A.lmax;
        formed.assign((size_t)2 * nbs * d_moffp(lmax, lmax + 1), 0.0);
        for (int bm = 0; bm < nmaps; ++bm)
            for (int m = 0; m <= lmax; ++m)
                for (int l = m; l <= lmax + 1; ++l)
                    for (int part = 0; part < 2; ++part)
                        formed[2 * ((d_moffp(lmax, m) + (l - m)) * nbs + bm) + part] = band_prep_part(
                            prep->comps, prep->ncomp, prep->sx, prep->w + (int64_t)bm * prep->ncomp * (lmax + 1),
                            prep->bm_stokes[bm], prep->cnorm, lmax, m, l, part,
                            prep->extra ? prep->extra + (int64_t)bm * (lmax + 1) * (lmax + 1) : nullptr);
        ast = formed.data();
    }
    const int nbmax = leg_max_batch(A.R);
    for (int k0 = 0; k0 < nmaps; k0 += nbmax) {
        const int nb = std::min(nbmax, nmaps - k0);
#define CMDR_S(RR, NN) case NN: synth_RN<RR, NN>(A, tasks, ntasks, ast, nbs, k0, ph, ph_stride); break;
        if (A.R == 1) {
            switch (nb) { CMDR_S(1, 1) CMDR_S(1, 2) CMDR_S(1, 3) CMDR_S(1, 4) CMDR_S(1, 5) CMDR_S(1, 6) CMDR_S(1, 7)
                          CMDR_S(1, 8) CMDR_S(1, 9) }
        } else if (A.R == 2) {
            switch (nb) { CMDR_S(2, 1) CMDR_S(2, 2) CMDR_S(2, 3) CMDR_S(2, 4) }
        } else {
            switch (nb) { CMDR_S(4, 1) CMDR_S(4, 2) CMDR_S(4, 3) }
        }
#undef CMDR_S
    }
}

template <int R, int NB, bool SQ>
static void adj_RN(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride, int k0,
                   double* part, int64_t pms, int64_t pcs) {
    const int lmax = A.lmax;
    for (int ti = 0; ti < ntasks; ++ti) {
        const WaveTask t = tasks[ti];
        if (t.chunk < 0) continue;
        std::vector<AdjLane<R, NB>> S(64);
        for (int lane = 0; lane < 64; ++lane) leg_adj_load<R, NB, SQ>(A, ph, ph_stride, k0, t.m, t.chunk, lane, S[lane]);
        const int64_t mo = d_moffp(lmax, t.m);
        const double* al = A.alpha + (mo - t.m);
        double* out0 = part + t.chunk * pcs + 2 * (mo - t.m);
        std::vector<double> wbuf((size_t)64 * kAdjL_ * R);
        for (int l0 = t.lw; l0 <= lmax; l0 += kAdjL_) {
            for (int lane = 0; lane < 64; ++lane) {
                double (*w)[R] = reinterpret_cast<double (*)[R]>(wbuf.data() + (size_t)lane * kAdjL_ * R);
                if (l0 < t.lAend) leg_adj_mu_group<R, NB, SQ, true>(al, l0, S[lane], w);
                else leg_adj_mu_group<R, NB, SQ, false>(al, l0, S[lane], w);
            }
            for (int k = 0; k < NB; ++k) {
                double wl[16 * 65];
                for (int lane = 0; lane < 64; ++lane) {
                    double v[16];
                    const double (*w)[R] = reinterpret_cast<const double (*)[R]>(wbuf.data() + (size_t)lane * kAdjL_ * R);
                    leg_adj_products<R, NB>(S[lane], w, k, v);
                    for (int j = 0; j < 16; ++j) wl[j * 65 + lane] = v[j];
                }
                for (int col = 0; col < 16; ++col) {  // device order: quarter sums, then xor-16 / xor-32 butterflies
                    double q[4];
                    for (int qt = 0; qt < 4; ++qt) {
                        double s = 0.0;
                        for (int i = 0; i < 16; ++i) s += wl[col * 65 + qt * 16 + i];
                        q[qt] = s;
                    }
                    const double s = (q[0] + q[1]) + (q[2] + q[3]);
                    const int l = l0 + (col >> 1);
                    if (l <= lmax) out0[(k0 + k) * pms + 2 * l + (col & 1)] = s;
                }
            }
        }
    }
}
void launch_leg_adj(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                    double* part, int64_t pms, int64_t pcs, int nmaps, bool square, hipStream_t,
                    const std::function<void(int)>& between) {
    if (between && !square) between(0);
    if (square) {
        for (int k0 = 0; k0 < nmaps; ++k0) {
            if (A.R == 1) adj_RN<1, 1, true>(A, tasks, ntasks, ph, ph_stride, k0, part, pms, pcs);
            else if (A.R == 2) adj_RN<2, 1, true>(A, tasks, ntasks, ph, ph_stride, k0, part, pms, pcs);
            else adj_RN<4, 1, true>(A, tasks, ntasks, ph, ph_stride, k0, part, pms, pcs);
        }
        return;
    }
    const int nbmax = leg_max_batch(A.R);
    for (int k0 = 0; k0 < nmaps; k0 += nbmax) {
        const int nb = std::min(nbmax, nmaps - k0);
#define CMDR_A(RR, NN) case NN: adj_RN<RR, NN, false>(A, tasks, ntasks, ph, ph_stride, k0, part, pms, pcs); break;
        if (A.R == 1) {
            switch (nb) { CMDR_A(1, 1) CMDR_A(1, 2) CMDR_A(1, 3) CMDR_A(1, 4) CMDR_A(1, 5) CMDR_A(1, 6) CMDR_A(1, 7)
                          CMDR_A(1, 8) CMDR_A(1, 9) }
        } else if (A.R == 2) {
            switch (nb) { CMDR_A(2, 1) CMDR_A(2, 2) CMDR_A(2, 3) CMDR_A(2, 4) }
        } else {
            switch (nb) { CMDR_A(4, 1) CMDR_A(4, 2) CMDR_A(4, 3) }
        }
#undef CMDR_A
    }
}

template <int R>
static void synth2_R(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* st, int npol, int ip, double* ph,
                     int64_t ph_stride, int kq) {
    for (int t = 0; t < ntasks; ++t)
        for (int lane = 0; lane < 64 && tasks[t].chunk >= 0; ++lane)
            leg2_synth_lane<R>(A, st, npol, ip, ph, ph_stride, kq, tasks[t].m, tasks[t].chunk, tasks[t].lw, tasks[t].lAend, lane);
}
void launch_leg2_synth(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* st, int npol, double* ph,
                       int64_t ph_stride, int kq0, hipStream_t) {
    for (int ip = 0; ip < npol; ++ip) {
        if (A.R == 1) synth2_R<1>(A, tasks, ntasks, st, npol, ip, ph, ph_stride, kq0 + 2 * ip);
        else synth2_R<2>(A, tasks, ntasks, st, npol, ip, ph, ph_stride, kq0 + 2 * ip);
    }
}
template <int R>
static void adj2_R(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride, int kq,
                   double* part, int64_t pcs) {
    const int lmax = A.lmax;
    for (int ti = 0; ti < ntasks; ++ti) {
        const WaveTask t = tasks[ti];
        if (t.chunk < 0) continue;
        std::vector<Leg2State<R>> S(64);
        std::vector<Adj2G<R>> G(64);
        for (int lane = 0; lane < 64; ++lane) {
            leg2_load_state<R>(A, t.m, t.chunk, lane, S[lane]);
            leg2_adj_load<R>(A, ph, ph_stride, kq, t.m, t.chunk, lane, G[lane]);
            for (int r = 0; r < R; ++r)
                if (S[lane].ls[r] == t.lw) {
                    S[lane].pc[r] = S[lane].sd(r, 0); S[lane].pp[r] = S[lane].sd(r, 1);
                    S[lane].mc[r] = S[lane].sd(r, 2); S[lane].mp[r] = S[lane].sd(r, 3);
                }
        }
        const int64_t mo = d_moffp(lmax, t.m);
        const double* al = A.alpha + (mo - t.m);
        const double* be = A.beta + (mo - t.m);
        double* out = part + t.chunk * pcs + 4 * (mo - t.m);
        for (int l0 = t.lw; l0 <= lmax; l0 += 4) {
            double wl[16 * 65];
            for (int lane = 0; lane < 64; ++lane) {
                double v[16];
                if (l0 < t.lAend) leg2_adj_group<R, true>(A, al, be, l0, S[lane], G[lane], v);
                else leg2_adj_group<R, false>(A, al, be, l0, S[lane], G[lane], v);
                for (int j = 0; j < 16; ++j) wl[j * 65 + lane] = v[j];
            }
            for (int col = 0; col < 16; ++col) {
                double q[4];
                for (int qt = 0; qt < 4; ++qt) {
                    double sacc = 0.0;
                    for (int i = 0; i < 16; ++i) sacc += wl[col * 65 + qt * 16 + i];
                    q[qt] = sacc;
                }
                const int l = l0 + (col >> 2);
                if (l <= lmax) out[4 * l + (col & 3)] = (q[0] + q[1]) + (q[2] + q[3]);
            }
        }
    }
}
void launch_leg2_adj(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                     int kq0, double* part, int64_t part_pol_stride, int64_t pcs, int npol, hipStream_t) {
    for (int ip = 0; ip < npol; ++ip) {
        if (A.R == 1) adj2_R<1>(A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip, part + ip * part_pol_stride, pcs);
        else adj2_R<2>(A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip, part + ip * part_pol_stride, pcs);
    }
}
void launch_alm2_to_stream(const double* aE, const double* aB, int64_t pol_stride, double* st, int npol,
                           const double* cnorm, int lmax, hipStream_t) {
    for (int ip = 0; ip < npol; ++ip)
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax + 1; ++l)
                alm2_to_stream_elem(aE + ip * pol_stride, aB + ip * pol_stride, st, npol, ip, cnorm, lmax, m, l);
}
void launch_part2_to_alm(const double* part, int64_t part_pol_stride, int64_t pcs, int nchunk, double* aE, double* aB,
                         int64_t pol_stride, const double* cnorm, int lmax, int npol, hipStream_t, const int* lwtab) {
    for (int ip = 0; ip < npol; ++ip)
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax; ++l)
                part2_to_alm_elem(part + ip * part_pol_stride, pcs, nchunk, aE + ip * pol_stride, aB + ip * pol_stride,
                                  cnorm, lmax, m, l, lwtab);
}

void launch_ring(int mode, const RingDev* rings, const int* cls, int ncls, int log2M, double* ph,
                 int64_t ph_stride, int64_t prow /* rows (m) per pair of the phase layout */, double* map, int64_t map_stride, const double* const* mul,
                 int weighted, const cd* tw, int log2Mmax, const cd* chirp, cd* scratch, int64_t scratch_map_stride,
                 int scratch_line, int nmaps, hipStream_t, const cd* that, int64_t that_stride) {
    std::vector<cd> bufv((size_t)lds_elems(log2M));
    cd* buf = bufv.data();
    const FftCtx c{0, 1};
    for (int imap = 0; imap < nmaps; ++imap)
        for (int ib = 0; ib < ncls; ++ib) {
            const int pair = cls[ib];
            const RingDev d = rings[pair];
            double* php = ph + imap * ph_stride;
            double* mp = map ? map + imap * map_stride : nullptr;
            const double* mu = mul ? mul[imap] : nullptr;
            const double wg = weighted ? d.wgt : 1.0;
            cd* sc = d.split ? scratch + imap * scratch_map_stride + (int64_t)(d.split - 1) * scratch_line : nullptr;
            if (mode == 0) ring_block<0>(buf, d, pair, php, prow, mp, mu, wg, tw, log2Mmax, chirp, sc, c);
            else if (mode == 1) ring_block<1>(buf, d, pair, php, prow, mp, mu, wg, tw, log2Mmax, chirp, sc, c);
            else ring_block<2>(buf, d, pair, php, prow, mp, mu, wg, tw, log2Mmax, chirp, sc, c,
                               that ? that + imap * that_stride : nullptr);
        }
}

void launch_ring_toeplitz_spec(const RingDev* rings, const int* cls, int ncls, int log2M, const double* td,
                               int64_t prow, cd* that, const cd* tw, int log2Mmax, hipStream_t) {
    std::vector<cd> bufv((size_t)lds_elems(log2M));
    for (int ib = 0; ib < ncls; ++ib) {
        const int pair = cls[ib];
        ring_toeplitz_spec(bufv.data(), rings[pair], td, prow, pair, that, tw, log2Mmax, FftCtx{0, 1});
    }
}

void launch_md_sums(const RingDev* rings, int npair, int nside, const double* map, const double* mask, int type,
                    double* out, hipStream_t) {
    for (int p = 0; p < npair; ++p) {
        const RingDev d = rings[p];
        double acc[kMdSums];
        for (int k = 0; k < kMdSums; ++k) acc[k] = 0.0;
        const double z = healpix_ring_z(nside, d.ring), sth = std::sqrt((1.0 - z) * (1.0 + z));
        const double dphi = 6.283185307179586476925287 / d.nphi;
        for (int k = 0; k < d.nphi; ++k) {
            const double phi = d.phi0 + dphi * k;
            md_pixel_accum(type, z, sth, phi, map[d.startN + k], mask[d.startN + k], acc);
            if (d.startS >= 0) md_pixel_accum(type, -z, sth, phi, map[d.startS + k], mask[d.startS + k], acc);
        }
        for (int k = 0; k < kMdSums; ++k) out[(int64_t)p * kMdSums + k] = acc[k];
    }
}
void launch_alm_to_stream(const double* alm, int64_t alm_stride, double* ast, const double* cnorm, int lmax,
                          int nmaps, hipStream_t) {
    for (int k = 0; k < nmaps; ++k)
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax + 1; ++l)
                alm_to_stream_elem(alm + k * alm_stride, ast, nmaps, k, cnorm, lmax, m, l);
}
void launch_part_to_alm(const double* part, int64_t pms, int64_t pcs, int nchunk, double* alm, int64_t alm_stride,
                        const double* cnorm, int lmax, int nmaps, hipStream_t, const int* lwtab) {
    for (int k = 0; k < nmaps; ++k)
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax; ++l)
                part_to_alm_elem(part + k * pms, pcs, nchunk, alm + k * alm_stride, cnorm, lmax, m, l, lwtab);
}

void launch_sqrtS(const CompDev* comps, int ncomp, int, const double* smat, int kind, const double* in,
                  const double* add, double* out, bool pass_inactive, hipStream_t) {
    for (int c = 0; c < ncomp; ++c)
        for (int m = 0; m <= comps[c].lmax; ++m)
            for (int l = m; l <= comps[c].lmax; ++l) sqrtS_elem(comps[c], smat, kind, in, add, out, m, l, pass_inactive);
}
void launch_band_prep(const CompDev* comps, int ncomp, const double* sx, const double* w, const int* bm_stokes,
                      double* ast, const double* cnorm, int lmax_g, int nbm, hipStream_t, const double* extra) {
    const int64_t na = (int64_t)(lmax_g + 1) * (lmax_g + 1);
    for (int bm = 0; bm < nbm; ++bm)
        for (int m = 0; m <= lmax_g; ++m)
            for (int l = m; l <= lmax_g + 1; ++l)
                band_prep_elem(comps, ncomp, sx, w + (int64_t)bm * ncomp * (lmax_g + 1), bm_stokes[bm], ast, nbm, bm,
                               cnorm, lmax_g, m, l, extra ? extra + bm * na : nullptr);
}
void launch_alm_copy_batch(const AlmCopyDesc* d, int n, hipStream_t) {
    for (int i = 0; i < n; ++i)
        for (int m = 0; m <= d[i].lmax_d; ++m)
            for (int l = m; l <= d[i].lmax_d; ++l)
                alm_copy_elem(d[i].src, d[i].lmax_s, d[i].dst, d[i].lmax_d, d[i].fl, d[i].accumulate, d[i].lcut, m, l);
}
void launch_alm_copy(const double* src, int lmax_s, double* dst, int lmax_d, const double* fl, bool accumulate,
                     hipStream_t, int lcut) {
    for (int m = 0; m <= lmax_d; ++m)
        for (int l = m; l <= lmax_d; ++l) alm_copy_elem(src, lmax_s, dst, lmax_d, fl, accumulate ? 1 : 0, lcut, m, l);
}
void launch_pinv_prior(const CompDev* comps, int ncomp, int, const double* Q, int lmax_pre, int nmaps_pre,
                       const double* x, const double* z, double* out, hipStream_t) {
    for (int m = 0; m <= lmax_pre; ++m)
        for (int l = m; l <= lmax_pre; ++l) pinv_prior_elem(comps, ncomp, Q, lmax_pre, nmaps_pre, x, z, out, m, l);
}
void launch_band_post(const CompDev* comps, int ncomp, int lmax_max, const double* part, int64_t pms, int64_t pcs,
                      int nchunk, int nbm, const int* bm_stokes, const double* w, const double* cnorm, int lmax_g,
                      double* yc, bool accumulate, hipStream_t, const int* lwtab, int m0, int m1) {
    if (m1 < 0 || m1 > lmax_max + 1) m1 = lmax_max + 1;
    for (int m = m0; m < m1; ++m)
        for (int l = m; l <= lmax_max; ++l)
            band_post_elem(comps, ncomp, part, pms, pcs, nchunk, nbm, bm_stokes, w, cnorm, lmax_g, yc, accumulate ? 1 : 0, m, l,
                           m <= lmax_g ? lwtab : nullptr);
}
void launch_band_prep2(const CompDev* comps, int ncomp, const double* sx, const double* w, int nT, double* st, int npol,
                       const double* cnorm2, int lmax_g, hipStream_t, const double* extra) {
    const int64_t ws = (int64_t)ncomp * (lmax_g + 1);
    const int64_t na = (int64_t)(lmax_g + 1) * (lmax_g + 1);
    for (int ip = 0; ip < npol; ++ip)
        for (int m = 0; m <= lmax_g; ++m)
            for (int l = m; l <= lmax_g + 1; ++l)
                band_prep2_elem(comps, ncomp, sx, w + (nT + 2 * ip) * ws, w + (nT + 2 * ip + 1) * ws, st, npol, ip, cnorm2,
                                lmax_g, m, l, extra ? extra + (nT + 2 * ip) * na : nullptr,
                                extra ? extra + (nT + 2 * ip + 1) * na : nullptr);
}
void launch_band_post2(const CompDev* comps, int ncomp, int, const double* part2, int64_t pps, int64_t pcs, int nchunk,
                       int npol, const double* w, int nT, const double* cnorm2, int lmax_g, double* yc, hipStream_t,
                       const int* lwtab) {
    for (int m = 0; m <= lmax_g; ++m)
        for (int l = m; l <= lmax_g; ++l)
            band_post2_elem(comps, ncomp, part2, pps, pcs, nchunk, npol, w, nT, cnorm2, lmax_g, yc, m, l, lwtab);
}
void launch_precond_diag(const CompDev* comps, int ncomp, const double* P, int lmax_pre, int nmaps_pre,
                         const double* in, double* out, hipStream_t) {
    for (int m = 0; m <= lmax_pre; ++m)
        for (int l = m; l <= lmax_pre; ++l) precond_diag_elem(comps, ncomp, P, lmax_pre, nmaps_pre, in, out, m, l);
}
void launch_fill_gl(double* ph, const double* wn, const double* ws, int npair_pad, int lmax, hipStream_t) {
    for (int m = 0; m <= lmax; ++m)
        for (int p = 0; p < npair_pad; ++p) {
            double* o = ph + d_phidx(lmax + 1, p, m);
            o[0] = wn[p]; o[1] = 0.0; o[2] = ws[p]; o[3] = 0.0;
        }
}
void launch_part_to_diag(const double* part, int64_t pcs, int nchunk, const double* cnorm, double* out, int lmax,
                         hipStream_t) {
    for (int m = 0; m <= lmax; ++m)
        for (int l = m; l <= lmax; ++l) {
            const int64_t t = d_moffp(lmax, m) + (l - m);
            double s = 0.0;
            for (int c = 0; c < nchunk; ++c) s += part[c * pcs + 2 * t];
            s *= cnorm[t] * cnorm[t];
            const int64_t i = d_packed_index(lmax, l, m);
            out[i] = s;
            if (m > 0) out[i + 1] = s;
        }
}
void launch_pix(int mode, const double* a, const double* b, const double* c, double* out, int64_t n, hipStream_t) {
    for (int64_t i = 0; i < n; ++i) out[i] = mode == 0 ? a[i] * b[i] : a[i] * (a[i] * b[i] + c[i]);
}
int dot_partial_count() { return 1; }
void launch_dot(const double* a, const double* b, int64_t n, double*, double* scal, int slot, bool shift,
                hipStream_t) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    if (shift) scal[slot + 1] = scal[slot];
    scal[slot] = s;
}
// fused PCG vector kernels: the partial-sum protocol with the whole sum in entry 0
static double fold_partials(const double* p) {
    double t = 0.0;
    for (int i = 0; i < dot_partial_count(); ++i) t += p[i];
    return t;
}
static void put_partial(double* p, double v) {
    std::fill(p, p + dot_partial_count(), 0.0);
    p[0] = v;
}
void launch_cg_seed(const double* scal, int slot, double* p, hipStream_t) { put_partial(p, scal[slot]); }
void launch_cg_q(const CompDev* comps, int ncomp, int lmax, const double* smat, const double* yc, const double* d,
                 double* q, double* p_dq, hipStream_t, int64_t ilo, int64_t ihi) {
    double acc = 0.0;
    for (int m = 0; m <= lmax; ++m)
        for (int l = m; l <= lmax; ++l)
            acc += cg_single(comps, ncomp) ? cg_q_elem1(comps[0], smat, yc, d, q, m, l, ilo, ihi) : cg_q_elem(comps, ncomp, smat, yc, d, q, m, l);
    put_partial(p_dq, acc);
}
void launch_cg_xr_precond(const CompDev* comps, int ncomp, int lmax, const double* P, int nmaps_pre, const double* p_dq,
                          const double* p_rs_old, double* p_rs, double* x, double* r, const double* d, const double* q,
                          double* sv, double* scal, hipStream_t, int64_t ilo, int64_t ihi) {
    const double dq = fold_partials(p_dq), alpha = fold_partials(p_rs_old) / dq;
    scal[2] = dq;
    double acc = 0.0;
    for (int m = 0; m <= lmax; ++m)
        for (int l = m; l <= lmax; ++l)
            acc += cg_single(comps, ncomp) && nmaps_pre == 1 ? cg_xr_elem1(comps[0], P, lmax, alpha, x, r, d, q, sv, m, l, ilo, ihi)
                                                             : cg_xr_elem(comps, ncomp, P, lmax, nmaps_pre, alpha, x, r, d, q, sv, m, l);
    put_partial(p_rs, acc);
}
void launch_cg_d_sqrtS(const CompDev* comps, int ncomp, int lmax, const double* smat, const double* p_rs_old,
                       const double* p_rs, double* d, const double* sv, double* sx, double* scal, hipStream_t, int64_t ilo,
                       int64_t ihi) {
    const double dold = fold_partials(p_rs_old), dnew = fold_partials(p_rs);
    scal[0] = dnew;
    scal[1] = dold;
    for (int m = 0; m <= lmax; ++m)
        for (int l = m; l <= lmax; ++l) {
            if (cg_single(comps, ncomp)) cg_d_elem1(comps[0], smat, dnew / dold, d, sv, sx, m, l, ilo, ihi);
            else cg_d_elem(comps, ncomp, smat, dnew / dold, d, sv, sx, m, l);
        }
}
void launch_cg_xr(double* x, double* r, const double* d, const double* q, int64_t n, const double* scal, int num,
                  int den, hipStream_t) {
    const double alpha = scal[num] / scal[den];
    for (int64_t i = 0; i < n; ++i) { x[i] += alpha * d[i]; r[i] -= alpha * q[i]; }
}
void launch_cg_d(double* d, const double* sv, int64_t n, const double* scal, int num, int den, hipStream_t) {
    const double beta = scal[num] / scal[den];
    for (int64_t i = 0; i < n; ++i) d[i] = sv[i] + beta * d[i];
}
void launch_index_copy(const double* src, const int64_t* idx, double* dst, int n, bool scatter, hipStream_t) {
    for (int i = 0; i < n; ++i) {
        if (scatter) dst[idx[i]] = src[i];
        else dst[i] = src[idx[i]];
    }
}
void launch_axpby(const double* a, const double* b, double cb, double* out, int64_t n, hipStream_t) {
    for (int64_t i = 0; i < n; ++i) out[i] = a[i] + cb * b[i];
}

void launch_compact_fwd(double* z, const CellBase& B, const int64_t* rows, const int64_t* ptr, const int* col,
                        const double* val, const double* a, int64_t nrows, hipStream_t) {
    for (int64_t r = 0; r < nrows; ++r) compact_fwd_row(z, B, rows, ptr, col, val, a, r);
}
void launch_compact_adj(const double* u, const CellBase& B, const int64_t* cptr, const int64_t* cell, const double* val,
                        const double* scale, double* y, int nparam, bool accumulate, double*, hipStream_t) {
    for (int p = 0; p < nparam; ++p) {
        double acc = 0.0;
        for (int64_t k = cptr[p]; k < cptr[p + 1]; ++k) acc += compact_adj_term(u, B, cell, val, k);
        const double v = acc * (scale ? scale[p] : 1.0);
        y[p] = accumulate ? y[p] + v : v;
    }
}
int compact_adj_scratch(int nparam) { return nparam; }
void launch_dense_mv(const double* M, const double* x, double* y, int n, hipStream_t) {
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int j = 0; j < n; ++j) acc += M[(int64_t)i * n + j] * x[j];
        y[i] = acc;
    }
}
void launch_vec_scale(int mode, const double* a, const double* sc, const double* b, const double* c, double* out, int n,
                      hipStream_t) {
    for (int i = 0; i < n; ++i) {
        double v = mode == 1 ? a[i] / sc[i] : a[i] * sc[i];
        if (mode >= 2 && b) v += b[i];
        if (mode == 3 && c) v += c[i] / sc[i];
        out[i] = v;
    }
}
void launch_phase_share(double* base, int64_t slot_stride, int n, int64_t elems, bool sum, hipStream_t) {
    for (int64_t i = 0; i < elems; ++i)
        for (int j = 1; j < n; ++j) {
            if (sum) base[i] += base[j * slot_stride + i];
            else base[j * slot_stride + i] = base[i];
        }
}
void launch_alm_chain(double* alm, int64_t alm_stride, float* c32, int lmax, int nmaps, bool to_chain, hipStream_t) {
    const int64_t na = (int64_t)(lmax + 1) * (lmax + 1);
    for (int k = 0; k < nmaps; ++k)
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax; ++l) alm_chain_elem(alm + k * alm_stride, c32 + k * na, lmax, to_chain ? 1 : 0, m, l);
}
void launch_sigma_l(const double* alm, int64_t stride, int lmax, int nmaps, double* out, hipStream_t) {
    const int nspec = nmaps * (nmaps + 1) / 2;
    for (int l = 0; l <= lmax; ++l) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int m = 0; m <= l; ++m) {
            const int64_t i0 = d_packed_index(lmax, l, m);
            for (int sl = 0; sl < (m == 0 ? 1 : 2); ++sl) {
                int k = 0;
                for (int a = 0; a < nmaps; ++a)
                    for (int b = a; b < nmaps; ++b) acc[k++] += alm[a * stride + i0 + sl] * alm[b * stride + i0 + sl];
            }
        }
        for (int k = 0; k < nspec; ++k) out[l + (int64_t)(lmax + 1) * k] = acc[k] / (double)(2 * l + 1);
    }
}

}  // namespace cmdr
