#include "cr_system.hpp"

#include <cmath>
#include <cstring>
#include <map>

namespace cmdr {

CrSystem::CrSystem(int device) {
    CMDR_HIP_CHECK(hipSetDevice(device));
    CMDR_HIP_CHECK(hipStreamCreate(&stream_));
    CMDR_HIP_CHECK(hipStreamCreate(&stream_ring_));
    CMDR_HIP_CHECK(hipStreamCreate(&stream_comm_));
    if (const char* e = std::getenv("CMDR_PIPELINE")) pipeline_ = std::atoi(e) != 0;
}

CrSystem::~CrSystem() {
    if (stream_) (void)hipStreamSynchronize(stream_);   // nothing of ours (kernels, RCCL operations) still queued
    for (Group& G : groups_) {
        for (hipEvent_t e : G.ev_synth) (void)hipEventDestroy(e);
        for (hipEvent_t e : G.ev_ring) (void)hipEventDestroy(e);
    }
    if (stream_ring_) (void)hipStreamDestroy(stream_ring_);
    if (stream_comm_) (void)hipStreamDestroy(stream_comm_);
    if (ev_half_) (void)hipEventDestroy(ev_half_);
    if (ev_comm_) (void)hipEventDestroy(ev_comm_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

bool CrSystem::pipelined(const Group& G) const {
    return pipeline_ && G.npol == 0 && G.nT > kPipeBatch && compacts_.empty();
}

void CrSystem::pipeline_events(Group& G, int nbatch) {
    while ((int)G.ev_synth.size() < nbatch) {
        hipEvent_t a, b;
        CMDR_HIP_CHECK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        CMDR_HIP_CHECK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        G.ev_synth.push_back(a);
        G.ev_ring.push_back(b);
    }
}

void CrSystem::sync() { CMDR_HIP_CHECK(hipStreamSynchronize(stream_)); }

void CrSystem::set_profile(bool on) {
    double ms[kProfKinds];
    long long n[kProfKinds];
    read_profile(ms, n, kProfKinds);
    for (int k = 0; k < kProfKinds; ++k) { prof_ms_[k] = 0; prof_n_[k] = 0; }
    profile_ = on;
}

void CrSystem::problem_info(int64_t* out) const {
    CMDR_REQUIRE(finalized_, "finalize first");
    const Group& G = groups_[0];
    const LegendreTables& L = G.plan->tables().leg;
    int64_t steps = 0;
    for (int m = 0; m <= L.lmax; ++m)
        for (int p = 0; p < L.npair; ++p) {
            const int v = L.ls[(size_t)m * L.npair_pad + p];
            if (v != kLsNever) steps += L.lmax - v + 1;
        }
    out[0] = G.nbm;
    out[1] = (int64_t)L.tasks.size();  // incl. padding slots
    out[2] = steps;
}

void CrSystem::problem_info_ext(int n, int64_t* out) const {
    CMDR_REQUIRE(n >= 3 && n <= 8, "bad size");
    problem_info(out);
    const Group& G = groups_[0];
    int64_t v[8] = {out[0], out[1], out[2], 0, G.nT, G.npol, G.plan->tables().leg.npair, (int64_t)groups_.size()};
    if (G.npol) {   // (ring pair, l, m) steps of one spin-2 launch for ONE polarisation pair
        const Legendre2Tables& L2 = G.plan->tables().leg2;
        for (int m = 0; m <= L2.lmax; ++m)
            for (int p = 0; p < G.plan->tables().leg.npair; ++p) {
                const int s = L2.ls[(size_t)m * L2.npair_pad + p];
                if (s != kLsNever) v[3] += L2.lmax - s + 1;
            }
    }
    for (int k = 0; k < n; ++k) out[k] = v[k];
}

void CrSystem::span_begin(int kind, hipStream_t st) {
    if (!profile_) return;
    Span s;
    s.kind = kind;
    s.s = st ? st : stream_;
    CMDR_HIP_CHECK(hipEventCreate(&s.a));
    CMDR_HIP_CHECK(hipEventCreate(&s.b));
    CMDR_HIP_CHECK(hipEventRecord(s.a, s.s));
    spans_.push_back(s);
    open_.push_back((int)spans_.size() - 1);
}

void CrSystem::span_end() {
    if (!profile_) return;
    CMDR_HIP_CHECK(hipEventRecord(spans_[open_.back()].b, spans_[open_.back()].s));
    open_.pop_back();
}

void CrSystem::read_profile(double* ms_sum, long long* count, int nkinds) {
    CMDR_REQUIRE(nkinds >= 1 && nkinds <= kProfKinds, "bad number of profile kinds");
    if (!spans_.empty()) {
        sync();
        CMDR_HIP_CHECK(hipStreamSynchronize(stream_ring_));
        for (Span& s : spans_) {
            float ms = 0.f;
            CMDR_HIP_CHECK(hipEventElapsedTime(&ms, s.a, s.b));
            prof_ms_[s.kind] += ms;
            prof_n_[s.kind] += 1;
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
        spans_.clear();
    }
    for (int k = 0; k < nkinds; ++k) { ms_sum[k] = prof_ms_[k]; count[k] = prof_n_[k]; }
}

void CrSystem::set_rings(int nside, const std::vector<int>& rings) {
    CMDR_REQUIRE(!finalized_, "set_rings after finalize");
    for (auto& rs : ring_sets_)
        if (rs.first == nside) { rs.second = rings; return; }
    ring_sets_.push_back({nside, rings});
}

int64_t CrSystem::band_npix(int b) const {
    const Band& B = bands_[b];
    if (B.group >= 0) return groups_[B.group].plan->npix_local();
    for (auto& rs : ring_sets_)
        if (rs.first == B.nside && !rs.second.empty()) {
            int64_t n = 0;
            for (int r : rs.second) n += (int64_t)healpix_ring(B.nside, r).nphi * (r == 2 * B.nside ? 1 : 2);
            return n;
        }
    return 12 * (int64_t)B.nside * B.nside;
}

int CrSystem::add_band(int nside, int lmax, int nmaps, const double* siN, const double* b_l, double mb_eff,
                       const double* sg_mask, const double* wring) {
    CMDR_REQUIRE(!finalized_, "add_band after finalize");
    CMDR_REQUIRE(comps_.empty(), "add all bands before the first component (F_mean is dimensioned by numband)");
    CMDR_REQUIRE(nmaps >= 1 && nmaps <= 3, "nmaps must be 1..3");
    CMDR_REQUIRE(siN && b_l, "siN / b_l is NULL");
    bands_.emplace_back();
    Band& B = bands_.back();
    B.nside = nside;
    B.lmax = lmax;
    B.nmaps = nmaps;
    B.mb_eff = mb_eff;
    B.b_l.assign(b_l, b_l + (size_t)(lmax + 1) * nmaps);
    if (wring) { B.wring.assign(wring, wring + 2 * nside); B.has_wring = true; }
    const int64_t np = band_npix((int)bands_.size() - 1) * nmaps;
    std::vector<double> mul(np);
    for (int64_t i = 0; i < np; ++i) mul[i] = siN[i] * siN[i] * (sg_mask ? sg_mask[i] : 1.0);  // comm_N_rms_mod.f90:264-273
    std::vector<double> s1(siN, siN + np);
    if (sg_mask) for (int64_t i = 0; i < np; ++i) s1[i] *= sg_mask[i];                         // :304-313 (applied twice = once, mask is 0/1)
    if (sg_mask) B.siN_raw.upload(siN, (size_t)np);
    B.Nmap_h.resize(np);
    for (int64_t i = 0; i < np; ++i)                                                            // :288-301
        B.Nmap_h[i] = siN[i] > 0.0 ? (sg_mask ? sg_mask[i] : 1.0) / (siN[i] * siN[i]) : 0.0;
    B.siN.upload(s1);
    B.mul.upload(mul);
    return (int)bands_.size() - 1;
}

int CrSystem::add_comp(int lmax_amp, int nmaps, int lmax_cl, const double* sqrtS, const double* sqrtInvS,
                       const double* S, const double* F_mean, int active) {
    CMDR_REQUIRE(!finalized_, "add_comp after finalize");
    CMDR_REQUIRE(!bands_.empty(), "add bands first");
    CMDR_REQUIRE(nmaps >= 1 && nmaps <= 3, "nmaps must be 1..3");
    CMDR_REQUIRE(comps_.size() < 8, "at most 8 diffuse components");
    CMDR_REQUIRE(F_mean, "F_mean is NULL");
    comps_.emplace_back();
    Comp& C = comps_.back();
    C.d.lmax = lmax_amp;
    C.d.nmaps = nmaps;
    C.d.nalm = nalm_packed(lmax_amp);
    C.d.lmax_cl = lmax_cl;
    C.d.active = active ? 1 : 0;
    C.d.pos = 0;
    C.d.smat_off = 0;
    if (lmax_cl >= 0) {
        CMDR_REQUIRE(sqrtS && sqrtInvS && S, "S tables are NULL");
        const size_t n = (size_t)nmaps * nmaps * (lmax_cl + 1);
        C.sqrtS.assign(sqrtS, sqrtS + n);
        C.sqrtInvS.assign(sqrtInvS, sqrtInvS + n);
        C.S.assign(S, S + n);
    }
    C.F_mean.assign(F_mean, F_mean + (size_t)bands_.size() * nmaps);
    C.F_map.resize(bands_.size());
    C.F_map_nm.assign(bands_.size(), 0);
    C.mulF.resize(bands_.size());
    C.mulF_dirty.assign(bands_.size(), 1);
    order_.push_back({0, (int)comps_.size() - 1});
    return (int)comps_.size() - 1;
}

void CrSystem::set_band_qucov(int band, const double* iN, const double* siN_mat) {
    CMDR_REQUIRE(!finalized_, "set_band_qucov after finalize");
    CMDR_REQUIRE(band >= 0 && band < (int)bands_.size() && iN && siN_mat, "bad arguments");
    Band& B = bands_[band];
    CMDR_REQUIRE(B.nmaps == 3, "a QU-covariance band has nmaps = 3");
    const int64_t np = band_npix(band);
    CMDR_REQUIRE(np == 12 * (int64_t)B.nside * B.nside, "QU-covariance bands are not ring-sharded (every rank holds all pixels)");
    const size_t n = (size_t)(2 * np) * (size_t)(2 * np);
    B.qucov_iN.upload(iN, n);
    B.qucov_siN.upload(siN_mat, n);
}

int CrSystem::add_compact(int nparam, const double* sigma, const double* mean, int active) {
    CMDR_REQUIRE(!finalized_, "add_compact after finalize");
    CMDR_REQUIRE(!bands_.empty(), "add bands first");
    CMDR_REQUIRE(nparam >= 1 && sigma && mean, "bad arguments");
    compacts_.emplace_back();
    Compact& K = compacts_.back();
    K.nparam = nparam;
    K.active = active ? 1 : 0;
    K.sigma.assign(sigma, sigma + nparam);
    K.mean.assign(mean, mean + nparam);
    for (double v : K.sigma) CMDR_REQUIRE(v > 0.0, "compact component: prior sigma must be > 0");
    order_.push_back({1, (int)compacts_.size() - 1});
    return (int)compacts_.size() - 1;
}

void CrSystem::set_compact_band(int block, int band, int64_t nnz, const int64_t* cell, const int* param,
                                const double* val) {
    CMDR_REQUIRE(!finalized_, "set_compact_band after finalize");
    CMDR_REQUIRE(block >= 0 && block < (int)compacts_.size() && band >= 0 && band < (int)bands_.size(), "bad block / band");
    CMDR_REQUIRE(nnz >= 0 && (nnz == 0 || (cell && param && val)), "bad arguments");
    Compact& K = compacts_[block];
    const int64_t ncell = band_npix(band) * bands_[band].nmaps;
    for (CompactBand& B : K.P) CMDR_REQUIRE(B.band != band, "compact band set twice");
    K.P.emplace_back();
    CompactBand& B = K.P.back();
    B.band = band;
    for (int64_t i = 0; i < nnz; ++i) {
        CMDR_REQUIRE(cell[i] >= 0 && cell[i] < ncell && param[i] >= 0 && param[i] < K.nparam, "compact entry out of range");
        B.h_cell.push_back(cell[i]);
        B.h_param.push_back(param[i]);
        B.h_val.push_back(val[i]);
    }
}

void CrSystem::finalize() {
    CMDR_REQUIRE(!finalized_, "finalize called twice");
    CMDR_REQUIRE(!bands_.empty() && !comps_.empty(), "need at least one band and one component");
    // stacked vector: comm_signal_mod.f90:113-125, comm_cr_mod.f90:467-501
    int64_t pos = 0;
    std::vector<double> smat;
    lmax_max_ = -1;
    for (auto& o : order_) {   // stacked-vector order = order of the add_comp / add_compact calls (compList order)
        if (o.first == 0) { comps_[o.second].d.pos = pos; pos += comps_[o.second].d.nalm * comps_[o.second].d.nmaps; }
        else { compacts_[o.second].pos = pos; pos += compacts_[o.second].nparam; }
    }
    for (Compact& K : compacts_) {
        compact_scratch_.ensure((size_t)compact_adj_scratch(K.nparam));
        K.sigma_dev.upload(K.sigma);
        K.mean_dev.upload(K.mean);
        for (CompactBand& B : K.P) {   // COO -> CSR over the touched cells (sorted) and CSC by parameter
            const size_t nnz = B.h_val.size();
            std::vector<size_t> idx(nnz);
            for (size_t i = 0; i < nnz; ++i) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return B.h_cell[a] < B.h_cell[b]; });
            std::vector<int64_t> rows, rptr;
            std::vector<int> rcol;
            std::vector<double> rval;
            for (size_t i : idx) {
                if (rows.empty() || rows.back() != B.h_cell[i]) { rows.push_back(B.h_cell[i]); rptr.push_back((int64_t)rcol.size()); }
                rcol.push_back(B.h_param[i]);
                rval.push_back(B.h_val[i]);
            }
            rptr.push_back((int64_t)rcol.size());
            B.nrows = (int64_t)rows.size();
            std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return B.h_param[a] < B.h_param[b]; });
            std::vector<int64_t> cptr(K.nparam + 1, 0), ccell;
            std::vector<double> cval;
            for (size_t i : idx) { cptr[B.h_param[i] + 1]++; ccell.push_back(B.h_cell[i]); cval.push_back(B.h_val[i]); }
            for (int p = 0; p < K.nparam; ++p) cptr[p + 1] += cptr[p];
            if (rows.empty()) { rows.push_back(0); }
            if (ccell.empty()) { ccell.push_back(0); cval.push_back(0.0); rcol.push_back(0); rval.push_back(0.0); }
            B.rows.upload(rows); B.rptr.upload(rptr); B.rcol.upload(rcol); B.rval.upload(rval);
            B.cptr.upload(cptr); B.ccell.upload(ccell); B.cval.upload(cval);
            B.h_cell.clear(); B.h_param.clear(); B.h_val.clear();
        }
    }
    for (Comp& C : comps_) {
        C.d.smat_off = (long long)smat.size();
        smat.insert(smat.end(), C.sqrtS.begin(), C.sqrtS.end());
        smat.insert(smat.end(), C.sqrtInvS.begin(), C.sqrtInvS.end());
        smat.insert(smat.end(), C.S.begin(), C.S.end());
        lmax_max_ = std::max(lmax_max_, C.d.lmax);
    }
    ncr_ = pos;
    std::vector<CompDev> cd;
    for (Comp& C : comps_) cd.push_back(C.d);
    comps_dev_.upload(cd);
    if (smat.empty()) smat.push_back(0.0);
    smat_.upload(smat);
    // groups of bands sharing one SHT plan
    const int ncomp = (int)comps_.size();
    for (int b = 0; b < (int)bands_.size(); ++b) {
        Band& B = bands_[b];
        int g = -1;
        for (int k = 0; k < (int)groups_.size(); ++k)
            if (groups_[k].nside == B.nside && groups_[k].lmax == B.lmax) {
                const Band& B0 = bands_[groups_[k].bands[0]];
                if (B0.has_wring == B.has_wring && B0.wring == B.wring) g = k;
            }
        if (g < 0) {
            groups_.emplace_back();
            g = (int)groups_.size() - 1;
            groups_[g].nside = B.nside;
            groups_[g].lmax = B.lmax;
        }
        Group& G = groups_[g];
        B.group = -1;  // assigned after the plan exists (band_npix uses it)
        G.bands.push_back(b);
        CMDR_REQUIRE(B.nmaps == 1 || B.nmaps == 3, "bands must have nmaps = 1 (T) or 3 (T,Q,U)");
    }
    for (Group& G : groups_) {   // map order: T of every band, then (Q,U) of every polarised band
        for (int b : G.bands) { G.bm_band.push_back(b); G.bm_stokes.push_back(0); }
        G.nT = (int)G.bands.size();
        for (int b : G.bands)
            if (bands_[b].nmaps == 3) {
                G.bm_band.push_back(b); G.bm_stokes.push_back(1);
                G.bm_band.push_back(b); G.bm_stokes.push_back(2);
                G.npol += 1;
            }
        G.nbm = G.nT + 2 * G.npol;
    }
    for (int g = 0; g < (int)groups_.size(); ++g) {
        Group& G = groups_[g];
        std::vector<int> rings;
        for (auto& rs : ring_sets_) if (rs.first == G.nside) rings = rs.second;
        const Band& B0 = bands_[G.bands[0]];
        G.plan = std::make_unique<ShtPlan>(G.nside, G.lmax, rings, B0.has_wring ? B0.wring.data() : nullptr, G.nbm,
                                           G.npol > 0);
        const int64_t np = G.plan->npix_local();
        std::vector<const double*> mp(G.nbm);
        std::vector<double> bl((size_t)G.nbm * (G.lmax + 1));
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            Band& B = bands_[b];
            CMDR_REQUIRE((int64_t)B.siN.size() == np * B.nmaps, "siN size does not match the plan's local map");
            mp[bm] = B.mul.get() + (int64_t)j * np;
            for (int l = 0; l <= G.lmax; ++l) bl[(size_t)bm * (G.lmax + 1) + l] = B.b_l[l + (size_t)(B.lmax + 1) * j] * B.mb_eff;
        }
        G.bl.upload(bl);
        G.bm_stokes_dev.upload(G.bm_stokes);
        G.mul_ptrs.upload(mp);
        G.plan->toeplitz_build(mp, G.that, stream_);       // N^-1 of the matvec in circulant form for the cap rings
        for (int b : G.bands) bands_[b].group = g;
    }
    (void)ncomp;
    rebuild_weights();
    rebuild_mixing();
    sx_.alloc(ncr_); yc_.alloc(ncr_); r_.alloc(ncr_); d_.alloc(ncr_); q_.alloc(ncr_); s_.alloc(ncr_); tmp_.alloc(ncr_);
    dot_partial_.alloc(dot_partial_count());
    scal_.alloc(16);
    scal_.zero(stream_);
    sync();
    finalized_ = true;
}

// w[bm][c][l] = F_mean * b_l * mb_eff for the components on the constant-mixing fast path; 0 for (band, component)
// pairs with a mixing map (they go through mix_forward / mix_adjoint), inactive components and l beyond either lmax.
void CrSystem::rebuild_weights() {
    const int ncomp = (int)comps_.size();
    for (Group& G : groups_) {
        std::vector<double> w((size_t)G.nbm * ncomp * (G.lmax + 1), 0.0);
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            const Band& B = bands_[b];
            for (int c = 0; c < ncomp; ++c) {
                const Comp& C = comps_[c];
                if (!C.d.active || j >= C.d.nmaps || !C.F_map[b].empty()) continue;
                const double F = C.F_mean[b + (size_t)bands_.size() * j];
                for (int l = 0; l <= std::min(G.lmax, C.d.lmax); ++l)
                    w[((size_t)bm * ncomp + c) * (G.lmax + 1) + l] = F * B.b_l[l + (size_t)(B.lmax + 1) * j] * B.mb_eff;
            }
        }
        G.w.upload(w);
        if (!literal_quirks_) { G.w_fwd.release(); continue; }
        // literal comm_cr_mod.f90:846-861: component c reads, above its own lmax, what the last earlier active component
        // of the list with lmax >= l (and enough Stokes columns) left in pmap%alm -- times c's own F_mean and the beam
        std::vector<double> wf = w;
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            const Band& B = bands_[b];
            for (int c = 0; c < ncomp; ++c) {
                const Comp& C = comps_[c];
                if (!C.d.active || j >= C.d.nmaps) continue;
                CMDR_REQUIRE(C.F_map[b].empty() || C.d.lmax >= G.lmax, "literal_quirks: components with a mixing map must reach the band's lmax");
                if (!C.F_map[b].empty()) continue;
                const double F = C.F_mean[b + (size_t)bands_.size() * j];
                for (int l = C.d.lmax + 1; l <= G.lmax; ++l) {
                    int src = -1;
                    for (int p = c - 1; p >= 0; --p)      // last writer of (l, column j) before c
                        if (comps_[p].d.active && comps_[p].d.lmax >= l && j < comps_[p].d.nmaps) { src = p; break; }
                    if (src < 0) continue;
                    wf[((size_t)bm * ncomp + src) * (G.lmax + 1) + l] += F * B.b_l[l + (size_t)(B.lmax + 1) * j] * B.mb_eff;
                }
            }
        }
        G.w_fwd.upload(wf);
    }
}

void CrSystem::set_literal_quirks(bool v) {
    literal_quirks_ = v;
    if (finalized_) { sync(); rebuild_weights(); }
}

void CrSystem::set_mixing_map(int comp, int band, const double* F, int nmaps) {
    CMDR_REQUIRE(comp >= 0 && comp < (int)comps_.size() && band >= 0 && band < (int)bands_.size(), "bad comp / band");
    Comp& C = comps_[comp];
    if (!F) {
        C.F_map[band].clear();
        C.F_map_nm[band] = 0;
    } else {
        const int nm = std::min(C.d.nmaps, bands_[band].nmaps);
        CMDR_REQUIRE(nmaps == nm, "mixing map must have min(component nmaps, band nmaps) columns");
        CMDR_REQUIRE(nm == 1 || nm == 3, "mixing maps need nmaps = 1 or 3");
        const int64_t np = band_npix(band);
        C.F_map[band].assign(F, F + np * nm);
        C.F_map_nm[band] = nm;
        C.mulF_dirty[band] = 1;
    }
    if (finalized_) {
        sync();
        rebuild_weights();
        rebuild_mixing();
    }
}

void CrSystem::set_cl_diag(int comp, const double* cl) {
    CMDR_REQUIRE(comp >= 0 && comp < (int)comps_.size(), "bad comp");
    Comp& C = comps_[comp];
    CMDR_REQUIRE(C.d.lmax_cl >= 0 && cl, "component has no C_l");
    C.cl_diag.assign(cl, cl + (size_t)(C.d.lmax_cl + 1) * C.d.nmaps);
}

// New S tables of one component after the C_l Gibbs step (sampleCls -> updateS, comm_Cl_mod.f90:838-863); the
// preconditioner is refreshed by the caller's next cmdr_precond_update_* as in the reference's Gibbs loop.
void CrSystem::set_comp_cl(int comp, const double* sqrtS, const double* sqrtInvS, const double* S) {
    CMDR_REQUIRE(comp >= 0 && comp < (int)comps_.size(), "bad comp");
    Comp& C = comps_[comp];
    CMDR_REQUIRE(C.d.lmax_cl >= 0, "component has no C_l (cltype none)");
    CMDR_REQUIRE(sqrtS && sqrtInvS && S, "S tables are NULL");
    const size_t n = (size_t)C.d.nmaps * C.d.nmaps * (C.d.lmax_cl + 1);
    C.sqrtS.assign(sqrtS, sqrtS + n);
    C.sqrtInvS.assign(sqrtInvS, sqrtInvS + n);
    C.S.assign(S, S + n);
    if (finalized_) {
        sync();
        double* d = smat_.get() + C.d.smat_off;
        CMDR_HIP_CHECK(hipMemcpy(d, C.sqrtS.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        CMDR_HIP_CHECK(hipMemcpy(d + n, C.sqrtInvS.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        CMDR_HIP_CHECK(hipMemcpy(d + 2 * n, C.S.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    }
}

void CrSystem::set_comp_f_mean(int comp, const double* F_mean) {
    CMDR_REQUIRE(comp >= 0 && comp < (int)comps_.size() && F_mean, "bad comp / F_mean");
    Comp& C = comps_[comp];
    C.F_mean.assign(F_mean, F_mean + (size_t)bands_.size() * C.d.nmaps);
    if (finalized_) {
        sync();
        rebuild_weights();
    }
}

// c%active_samp_group(samp_group) of a diffuse (kind 0) or compact (kind 1) component
void CrSystem::set_active(int kind, int idx, int active) {
    if (kind == 0) {
        CMDR_REQUIRE(idx >= 0 && idx < (int)comps_.size(), "bad comp");
        comps_[idx].d.active = active ? 1 : 0;
    } else {
        CMDR_REQUIRE(idx >= 0 && idx < (int)compacts_.size(), "bad compact block");
        compacts_[idx].active = active ? 1 : 0;
    }
    if (finalized_) {
        sync();
        std::vector<CompDev> cd;
        for (Comp& C : comps_) cd.push_back(C.d);
        comps_dev_.upload(cd);
        rebuild_weights();
        rebuild_mixing();
    }
}

// Batches of (band, component) pairs with a mixing map, per plan: scalar columns first, then (Q,U) pairs, as many
// per sandwich() call as the plan has map slots.
void CrSystem::rebuild_mixing() {
    for (Group& G : groups_) {
        ShtPlan& P = *G.plan;
        const int64_t np = P.npix_local(), na = P.nalm();
        const std::vector<double> pw = P.pixel_weights();
        G.mix.clear();
        const int cap = P.max_maps();
        bool any = false;
        for (int c = 0; c < (int)comps_.size(); ++c) {   // a batch holds columns of ONE component: they share its synthesis
            Comp& C = comps_[c];
            std::vector<MixCol> T, Pp;
            for (int ib = 0; ib < (int)G.bands.size(); ++ib) {
                const int b = G.bands[ib];
                if (C.F_map[b].empty()) { C.mulF[b] = DevBuf<double>(); continue; }
                const int nm = C.F_map_nm[b];
                if (C.mulF_dirty[b] || C.mulF[b].size() != (size_t)np * nm) {   // a change of sampling group keeps the maps
                    std::vector<double> mf((size_t)np * nm);
                    for (int j = 0; j < nm; ++j)
                        for (int64_t i = 0; i < np; ++i) mf[(size_t)j * np + i] = C.F_map[b][(size_t)j * np + i] * pw[i];
                    C.mulF[b].upload(mf);
                    C.mulF_dirty[b] = 0;
                }
                if (!C.d.active) continue;                      // comm_cr_mod.f90:851-854
                T.push_back({ib, c, 0});
                if (nm == 3) {
                    int ip = 0;
                    for (int k = 0; k < ib; ++k) if (bands_[G.bands[k]].nmaps == 3) ++ip;
                    Pp.push_back({G.nT + 2 * ip, c, 1});
                }
            }
            size_t it = 0, ip = 0;
            while (it < T.size() || ip < Pp.size()) {
                MixBatch B;
                int used = 0;
                while (it < T.size() && used < cap) { B.T.push_back(T[it++]); ++used; }
                while (ip < Pp.size() && used + 2 <= cap) { B.P.push_back(Pp[ip++]); used += 2; }
                CMDR_REQUIRE(used > 0, "plan has too few map slots for a polarised mixing pair");
                std::vector<const double*> mp;
                for (const MixCol& m : B.T) mp.push_back(comps_[m.comp].mulF[G.bm_band[m.bm]].get());
                for (const MixCol& m : B.P) {
                    const double* f = comps_[m.comp].mulF[G.bm_band[m.bm]].get();
                    mp.push_back(f + np);
                    mp.push_back(f + 2 * np);
                }
                B.mul_ptrs.upload(mp);
                G.mix.push_back(std::move(B));
                any = true;
            }
        }
        if (!any) continue;
        G.mix_in.ensure((size_t)cap * na);
        G.mix_out.ensure((size_t)cap * na);
        G.E.ensure((size_t)G.nbm * na);
        G.U.ensure((size_t)G.nbm * na);
    }
}

// CMDR_MIX_SHARE=0: every (band, component) pair runs its own synthesis and adjoint (development A/B switch)
static bool mix_share() {
    static int v = -1;
    if (v < 0) { v = 1; if (const char* e = std::getenv("CMDR_MIX_SHARE")) v = std::atoi(e) != 0; }
    return v == 1;
}

// E[bm] = b_l * sum_{c with a mixing map} YtW F_bc Y (S^1/2 x)_c      (evalDiffuseBand, :2082-2089)
void CrSystem::mix_forward(Group& G, const double* sx) {
    ShtPlan& P = *G.plan;
    const int64_t na = P.nalm();
    CMDR_HIP_CHECK(hipMemsetAsync(G.E.get(), 0, sizeof(double) * G.nbm * na, stream_));
    for (MixBatch& B : G.mix) {
        const int nT = (int)B.T.size(), nP = (int)B.P.size();
        int k = 0;
        std::vector<AlmCopyDesc> cp;     // the staging copies of a batch go out as one launch
        auto col_in = [&](const MixCol& m, int stokes) {
            const CompDev& C = comps_[m.comp].d;
            cp.push_back({sx + C.pos + (int64_t)stokes * C.nalm, G.mix_in.get() + (int64_t)k * na, nullptr, C.lmax, G.lmax,
                          0, 1 << 30});
            ++k;
        };
        // every column of a batch starts from the same component a_lm: with sharing it is staged (and synthesised) once
        const bool share = mix_share();
        for (const MixCol& m : B.T) { col_in(m, 0); if (share) break; }
        for (const MixCol& m : B.P) { col_in(m, 1); col_in(m, 2); if (share) break; }
        launch_alm_copy_batch(cp.data(), (int)cp.size(), stream_);
        P.sandwich(G.mix_in.get(), G.mix_out.get(), B.mul_ptrs.get(), nT, nP, stream_, share, false);
        reduce_rings(G.mix_out.get(), (int64_t)(nT + 2 * nP) * na);
        k = 0;
        cp.clear();
        auto col_out = [&](int bm) {     // one component's columns: every band map of the batch is a different E[bm]
            cp.push_back({G.mix_out.get() + (int64_t)k * na, G.E.get() + (int64_t)bm * na,
                          G.bl.get() + (int64_t)bm * (G.lmax + 1), G.lmax, G.lmax, 1, 1 << 30});
            ++k;
        };
        for (const MixCol& m : B.T) col_out(m.bm);
        for (const MixCol& m : B.P) { col_out(m.bm); col_out(m.bm + 1); }
        launch_alm_copy_batch(cp.data(), (int)cp.size(), stream_);
    }
}

// yc_c += YtW F_bc Y b_l U[bm]  for the pairs with a mixing map          (projectDiffuseBand, :2153-2158)
// rhs: cr_computeRHS first cuts the band a_lm to the component's lmax (alm_equal, comm_cr_mod.f90:634) and mixes
// after that (:640-650); cr_matmulA mixes at the band's lmax and cuts afterwards (projectDiffuseBand).
void CrSystem::mix_adjoint(Group& G, bool rhs) {
    ShtPlan& P = *G.plan;
    const int64_t na = P.nalm();
    launch_part_to_alm(P.partials(), P.part_map_stride(), P.leg().tri_elems(), P.leg().nchunk, G.U.get(), na,
                       P.leg().cnorm.get(), G.lmax, G.nT, stream_, P.leg().lw_chunk.get());
    if (G.npol)
        launch_part2_to_alm(P.partials2(), P.part2_pol_stride(), P.leg2().tri4(), P.leg2().nchunk,
                            G.U.get() + (int64_t)G.nT * na, G.U.get() + (int64_t)(G.nT + 1) * na, 2 * na,
                            P.leg2().cnorm.get(), G.lmax, G.npol, stream_, P.leg2().lw_chunk.get());
    reduce_rings(G.U.get(), (int64_t)G.nbm * na);
    for (MixBatch& B : G.mix) {
        const int nT = (int)B.T.size(), nP = (int)B.P.size();
        int k = 0;
        std::vector<AlmCopyDesc> cp;
        auto col_in = [&](int bm, int comp) {
            cp.push_back({G.U.get() + (int64_t)bm * na, G.mix_in.get() + (int64_t)k * na,
                          G.bl.get() + (int64_t)bm * (G.lmax + 1), G.lmax, G.lmax, 0,
                          rhs ? comps_[comp].d.lmax : (1 << 30)});
            ++k;
        };
        for (const MixCol& m : B.T) col_in(m.bm, m.comp);
        for (const MixCol& m : B.P) { col_in(m.bm, m.comp); col_in(m.bm + 1, m.comp); }
        launch_alm_copy_batch(cp.data(), (int)cp.size(), stream_);
        // ... and every output is added to the same component block: with sharing the phases are summed and one adjoint runs
        const bool share = mix_share();
        P.sandwich(G.mix_in.get(), G.mix_out.get(), B.mul_ptrs.get(), nT, nP, stream_, false, share);
        k = 0;
        auto col_out = [&](const MixCol& m, int stokes) {
            const CompDev& C = comps_[m.comp].d;
            launch_alm_copy(G.mix_out.get() + (int64_t)k * na, G.lmax, yc_.get() + C.pos + (int64_t)stokes * C.nalm, C.lmax,
                            nullptr, true, stream_);
            ++k;
        };
        for (const MixCol& m : B.T) { col_out(m, 0); if (share) break; }
        for (const MixCol& m : B.P) { col_out(m, 1); col_out(m, 2); if (share) break; }
    }
}

// ------------------------------------------------------------------------------------------------- compact components
void CrSystem::qucov_invN(Group& G, int band, double* maps) {   // matmulInvN_1map, comm_N_QUcov_mod.f90:320-340
    const CellBase cb = cell_base(band);
    const int n = (int)(2 * cb.np);
    qucov_tmp_.ensure((size_t)n);
    CMDR_HIP_CHECK(hipMemsetAsync(maps + cb.off[0], 0, sizeof(double) * cb.np, stream_));
    launch_dense_mv(bands_[band].qucov_iN.get(), maps + cb.off[1], qucov_tmp_.get(), n, stream_);   // Q, U maps are adjacent
    CMDR_HIP_CHECK(hipMemcpyAsync(maps + cb.off[1], qucov_tmp_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
}

bool CrSystem::group_has_compact(const Group& G) const {
    for (int b : G.bands)
        if (bands_[b].qucov_iN.size()) return true;
    for (const Compact& K : compacts_)
        if (K.active)
            for (const CompactBand& B : K.P)
                if (bands_[B.band].group >= 0 && &groups_[bands_[B.band].group] == &G) return true;
    return false;
}

CellBase CrSystem::cell_base(int band) const {   // where the band's Stokes maps sit in its plan's [nbm][npix] buffers
    const Group& G = groups_[bands_[band].group];
    CellBase C;
    C.np = G.plan->npix_local();
    C.off[0] = C.off[1] = C.off[2] = 0;
    for (int bm = 0; bm < G.nbm; ++bm)
        if (G.bm_band[bm] == band) C.off[G.bm_stokes[bm]] = (int64_t)bm * C.np;
    return C;
}

void CrSystem::compact_forward(Group& G, const double* sx, double* maps) {   // evalPtsrcBand / evalTemplateBand
    for (Compact& K : compacts_) {
        if (!K.active) continue;
        for (CompactBand& B : K.P) {
            if (&groups_[bands_[B.band].group] != &G) continue;
            launch_compact_fwd(maps, cell_base(B.band), B.rows.get(), B.rptr.get(), B.rcol.get(), B.rval.get(), sx + K.pos,
                               B.nrows, stream_);
        }
    }
}

void CrSystem::compact_adjoint(Group& G, const double* maps, double* yc) {   // projectPtsrcBand / projectTemplateBand
    for (Compact& K : compacts_) {
        if (!K.active) continue;
        for (CompactBand& B : K.P) {
            if (&groups_[bands_[B.band].group] != &G) continue;
            launch_compact_adj(maps, cell_base(B.band), B.cptr.get(), B.ccell.get(), B.cval.get(), nullptr, yc + K.pos,
                               K.nparam, true, compact_scratch_.get(), stream_);
        }
    }
}

// Dense block of A on every compact block, inverted on the host: delta + sigma (sum_b P_b^t N_b^-1 P_b) sigma.
// (initPtsrcPrecond / initTemplatePrecond build approximations of this block; the exact one is used here.)
void CrSystem::compact_precond_init() {
    for (Compact& K : compacts_) {
        const int n = K.nparam;
        std::vector<double> M((size_t)n * n, 0.0);
        if (K.active) {
            DevBuf<double> e(n), col(n);
            std::vector<double> he(n, 0.0), hc(n);
            for (int j = 0; j < n; ++j) {
                he.assign(n, 0.0);
                he[j] = 1.0;
                e.upload(he, stream_);
                CMDR_HIP_CHECK(hipMemsetAsync(col.get(), 0, sizeof(double) * n, stream_));
                for (CompactBand& B : K.P) {
                    Group& G = groups_[bands_[B.band].group];
                    const int64_t np = G.plan->npix_local();
                    G.tmpmap.ensure((size_t)G.nbm * np);
                    const CellBase cb = cell_base(B.band);
                    for (int st = 0; st < bands_[B.band].nmaps; ++st)
                        CMDR_HIP_CHECK(hipMemsetAsync(G.tmpmap.get() + cb.off[st], 0, sizeof(double) * np, stream_));
                    launch_compact_fwd(G.tmpmap.get(), cb, B.rows.get(), B.rptr.get(), B.rcol.get(), B.rval.get(), e.get(),
                                       B.nrows, stream_);
                    for (int st = 0; st < bands_[B.band].nmaps; ++st)
                        launch_pix(0, bands_[B.band].mul.get() + (int64_t)st * np, G.tmpmap.get() + cb.off[st], nullptr,
                                   G.tmpmap.get() + cb.off[st], np, stream_);
                    launch_compact_adj(G.tmpmap.get(), cb, B.cptr.get(), B.ccell.get(), B.cval.get(), nullptr, col.get(), n,
                                       true, compact_scratch_.get(), stream_);
                }
                sync();
                CMDR_HIP_CHECK(hipMemcpy(hc.data(), col.get(), sizeof(double) * n, hipMemcpyDeviceToHost));
                for (int i = 0; i < n; ++i) M[(size_t)i * n + j] = hc[i];
            }
            DevBuf<double> dm(M.size());                        // sum over ring sets and band groups
            dm.upload(M, stream_);
            reduce(dm.get(), (int64_t)M.size());
            sync();
            CMDR_HIP_CHECK(hipMemcpy(M.data(), dm.get(), sizeof(double) * M.size(), hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) M[(size_t)i * n + j] *= K.sigma[i] * K.sigma[j];
        }
        for (int i = 0; i < n; ++i) M[(size_t)i * n + i] += 1.0;
        // Gauss-Jordan inverse (SPD block, small)
        std::vector<double> I((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0;
        for (int c = 0; c < n; ++c) {
            int piv = c;
            for (int r = c + 1; r < n; ++r) if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c])) piv = r;
            CMDR_REQUIRE(M[(size_t)piv * n + c] != 0.0, "singular compact preconditioner block");
            if (piv != c) for (int k = 0; k < n; ++k) { std::swap(M[(size_t)c * n + k], M[(size_t)piv * n + k]); std::swap(I[(size_t)c * n + k], I[(size_t)piv * n + k]); }
            const double inv = 1.0 / M[(size_t)c * n + c];
            for (int k = 0; k < n; ++k) { M[(size_t)c * n + k] *= inv; I[(size_t)c * n + k] *= inv; }
            for (int r = 0; r < n; ++r) {
                if (r == c) continue;
                const double f = M[(size_t)r * n + c];
                if (f == 0.0) continue;
                for (int k = 0; k < n; ++k) { M[(size_t)r * n + k] -= f * M[(size_t)c * n + k]; I[(size_t)r * n + k] -= f * I[(size_t)c * n + k]; }
            }
        }
        K.Minv.upload(I, stream_);
    }
}

void CrSystem::init_rccl(const char* id, int rank, int nranks) {
    CMDR_REQUIRE(id, "id is NULL");
    rccl_.init(id, rank, nranks);
}

void CrSystem::rccl_split_rings(int band_group, int ring_index, int ring_replicas) {
    CMDR_REQUIRE(rccl_.ready(), "cmdr_ctx_init_rccl first");
    CMDR_REQUIRE(ring_replicas >= 1 && ring_index >= 0 && ring_index < ring_replicas, "bad ring group");
    rccl_rings_.split_from(rccl_, band_group, ring_index);       // collective: every rank of the world calls it
    CMDR_REQUIRE(rccl_rings_.size() == ring_replicas, "ring group size does not match the communicator split");
    ring_replicas_ = ring_replicas;
    band_sharded_ = true;
}

// Back to the callbacks: bench.py's "all ranks fall back together" needs a rank whose own init succeeded to stop using
// its half-built communicators when another rank failed (reduce / reduce_rings give the native ones precedence).
void CrSystem::drop_rccl() {
    sync();
    const bool split = rccl_rings_.ready();
    rccl_rings_.destroy();
    rccl_.destroy();
    if (split) { band_sharded_ = false; ring_replicas_ = 1; }   // set by rccl_split_rings; cmdr_ctx_set_band_sharding sets them again
}

void CrSystem::reduce_rings(double* v, int64_t n) {
    if (!band_sharded_) { reduce(v, n); return; }
    if (rccl_rings_.ready()) {
        if (ring_replicas_ > 1) rccl_rings_.allreduce_sum(v, n, reinterpret_cast<void*>(stream_));
        return;
    }
    if (!allreduce_rings_) return;           // one rank per ring group
    sync();
    allreduce_rings_(allreduce_rings_user_, v, n);
}

void CrSystem::reduce(double* v, int64_t n) {
    if (rccl_.ready()) {  // RCCL on the library stream: stays queued behind the kernels that produced v
        rccl_.allreduce_sum(v, n, reinterpret_cast<void*>(stream_));
        return;
    }
    if (allreduce_s_) {   // stream-ordered collective: stays queued behind the kernels that produced v
        allreduce_s_(allreduce_s_user_, v, n, reinterpret_cast<void*>(stream_));
        return;
    }
    if (!allreduce_) return;
    sync();
    allreduce_(allreduce_user_, v, n);
}

void CrSystem::set_vector_slicing(int rank, int nranks) {
    CMDR_REQUIRE(nranks <= 1 || (rank >= 0 && rank < nranks), "bad rank / nranks");
    slice_rank_ = rank;
    slice_n_ = nranks > 1 ? nranks : 0;
    // test hook: a communicator of ONE rank still goes through ncclReduceScatter / ncclAllGather (tests/test_rccl_gpu.py)
    if (nranks == 1 && std::getenv("CMDR_SLICE_FORCE")) slice_n_ = 1;
}

// Reduce-scatter / all-gather of a padded stacked vector (slice_n_ chunks of slice_count() doubles), in place, on the
// library stream.  Native RCCL: ncclReduceScatter / ncclAllGather.  Callback drivers (MPI, the gloo tests) only have an
// all-reduce: the same result through it (reduce-scatter = all-reduce; all-gather = all-reduce of the vector with the
// chunks of the other ranks zeroed) -- correct, without the traffic saving.
void CrSystem::slice_reduce_scatter(double* v) {
    const int64_t c = slice_count();
    if (rccl_.ready()) { rccl_.reduce_scatter_sum(v, c, reinterpret_cast<void*>(stream_)); return; }
    reduce(v, c * slice_n_);
}
void CrSystem::slice_all_gather(double* v) {
    const int64_t c = slice_count();
    if (rccl_.ready()) { rccl_.all_gather(v, c, reinterpret_cast<void*>(stream_)); return; }
    if (slice_rank_ > 0) CMDR_HIP_CHECK(hipMemsetAsync(v, 0, sizeof(double) * c * slice_rank_, stream_));
    if (slice_rank_ + 1 < slice_n_)
        CMDR_HIP_CHECK(hipMemsetAsync(v + c * (slice_rank_ + 1), 0, sizeof(double) * c * (slice_n_ - slice_rank_ - 1), stream_));
    reduce(v, c * slice_n_);
}

// Sum over ranks of the rows m0 <= m < m1 of every diffuse block of a stacked vector, stream-ordered on `st` (native RCCL
// or the stream callback; one group call for all blocks).
void CrSystem::reduce_rows(double* v, int m0, int m1, hipStream_t st) {
    struct Rng { int64_t off, n; };
    std::vector<Rng> rs;
    for (const Comp& C : comps_) {
        const int lm = C.d.lmax;
        const int a = std::min(m0, lm + 1), b = std::min(m1, lm + 1);
        if (b <= a) continue;
        const int64_t i0 = a > lm ? C.d.nalm : d_packed_index(lm, a, a), i1 = b > lm ? C.d.nalm : d_packed_index(lm, b, b);
        for (int j = 0; j < C.d.nmaps; ++j) rs.push_back({C.d.pos + (int64_t)j * C.d.nalm + i0, i1 - i0});
    }
    if (rs.empty()) return;
    if (rccl_.ready()) {
        if (rs.size() > 1) RcclComm::group_start();
        for (const Rng& r : rs) rccl_.allreduce_sum(v + r.off, r.n, reinterpret_cast<void*>(st));
        if (rs.size() > 1) RcclComm::group_end();
        return;
    }
    for (const Rng& r : rs) allreduce_s_(allreduce_s_user_, v + r.off, r.n, reinterpret_cast<void*>(st));
}

// ------------------------------------------------------------------------------------------------- matvec
void CrSystem::adjoint_groups_to_yc(bool from_maps) {
    const int ncomp = (int)comps_.size();
    // Several ranks, stream-ordered collective, one unpolarised plan without mixing operators (the benchmark's layout):
    // the adjoint runs as two launches over m < m_split | m >= m_split (equal work), and the sum over ranks of the first
    // half's rows goes out on a second stream while the second half computes -- no host synchronisation, same sums in the
    // same order as the one-piece form (bit-equal results).  CMDR_OVERLAP=0 keeps one launch + one all-reduce.
    const bool overlap_env = [] { const char* e = std::getenv("CMDR_OVERLAP"); return !e || std::atoi(e) != 0; }();
    if (slice_active_) {   // sliced PCG loop (one unpolarised plan): every rank needs only its chunk of the sum
        Group& G = groups_[0];
        ShtPlan& P = *G.plan;
        span_begin(2);
        P.adjoint_to_partials(G.nT, false, stream_, nullptr);
        span_end();
        launch_band_post(comps_dev_.get(), ncomp, lmax_max_, P.partials(), P.part_map_stride(), P.leg().tri_elems(),
                         P.leg().nchunk, G.nT, G.bm_stokes_dev.get(), G.w.get(), P.leg().cnorm.get(), G.lmax, yc_.get(), false,
                         stream_, P.leg().lw_chunk.get());
        slice_reduce_scatter(yc_.get());
        return;
    }
    if (overlap_env && (rccl_.ready() || allreduce_s_) && !band_sharded_ && groups_.size() == 1 && groups_[0].npol == 0 &&
        groups_[0].mix.empty() && !groups_[0].ring_pending && compacts_.empty() && !group_has_compact(groups_[0])) {
        Group& G = groups_[0];
        ShtPlan& P = *G.plan;
        const int ms = P.m_split();
        if (!ev_half_) {
            CMDR_HIP_CHECK(hipEventCreateWithFlags(&ev_half_, hipEventDisableTiming));
            CMDR_HIP_CHECK(hipEventCreateWithFlags(&ev_comm_, hipEventDisableTiming));
        }
        span_begin(2);
        for (int half = 0; half < 2; ++half) {
            P.adjoint_half_to_partials(G.nT, half, stream_);
            launch_band_post(comps_dev_.get(), ncomp, lmax_max_, P.partials(), P.part_map_stride(), P.leg().tri_elems(),
                             P.leg().nchunk, G.nT, G.bm_stokes_dev.get(), G.w.get(), P.leg().cnorm.get(), G.lmax, yc_.get(),
                             false, stream_, P.leg().lw_chunk.get(), half == 0 ? 0 : ms, half == 0 ? ms : -1);
            if (half == 0) {
                CMDR_HIP_CHECK(hipEventRecord(ev_half_, stream_));
                CMDR_HIP_CHECK(hipStreamWaitEvent(stream_comm_, ev_half_, 0));
                reduce_rows(yc_.get(), 0, ms, stream_comm_);
                CMDR_HIP_CHECK(hipEventRecord(ev_comm_, stream_comm_));
            } else {
                reduce_rows(yc_.get(), ms, lmax_max_ + 1, stream_);
                CMDR_HIP_CHECK(hipStreamWaitEvent(stream_, ev_comm_, 0));
            }
        }
        span_end();
        return;
    }
    for (int g = 0; g < (int)groups_.size(); ++g) {
        Group& G = groups_[g];
        ShtPlan& P = *G.plan;
        span_begin(2);
        if (G.ring_pending) {
            const int nbatch = (G.nT + kPipeBatch - 1) / kPipeBatch;
            for (int j = 0; j < nbatch; ++j) {
                const int k0 = j * kPipeBatch, nb = std::min(kPipeBatch, G.nT - k0);
                CMDR_HIP_CHECK(hipStreamWaitEvent(stream_, G.ev_ring[j], 0));
                P.adjoint_range(k0, nb, stream_);
            }
            G.ring_pending = false;
        } else {
            // the matrix-unit launch(es) and the VALU launch(es) of the remaining maps are timed on their own (kinds 4, 5)
            span_begin(4);
            P.adjoint_to_partials(G.nT, false, stream_, [&](int nmx) {
                if (nmx > 0) span_end();
                else if (profile_) { spans_[open_.back()].kind = 5; return; }
                span_begin(5);
            });
            span_end();
            if (G.npol) { span_begin(7); P.adjoint2_to_partials(G.npol, G.nT, stream_); span_end(); }
        }
        span_end();
        launch_band_post(comps_dev_.get(), ncomp, lmax_max_, P.partials(), P.part_map_stride(), P.leg().tri_elems(),
                         P.leg().nchunk, G.nT, G.bm_stokes_dev.get(), G.w.get(), P.leg().cnorm.get(), G.lmax,
                         yc_.get(), g > 0, stream_, P.leg().lw_chunk.get());
        if (G.npol)
            launch_band_post2(comps_dev_.get(), ncomp, lmax_max_, P.partials2(), P.part2_pol_stride(), P.leg2().tri4(),
                              P.leg2().nchunk, G.npol, G.w.get(), G.nT, P.leg2().cnorm.get(), G.lmax, yc_.get(),
                              stream_, P.leg2().lw_chunk.get());
        if (!G.mix.empty()) mix_adjoint(G, from_maps);
    }
    reduce(yc_.get(), ncr_);
}

void CrSystem::matmulA(const double* x, double* y) { matmulA_impl(x, y, false, true); }

// sx_ready: sx_ already holds S^1/2 x (written by the fused PCG update of d).  finish = false: stop after the reduced
// yc_ = sum_nu F^t B^t Y^t N^-1 Y B F sx; the caller forms S^1/2 yc + x itself (k_cg_q, fused with the d.q product).
void CrSystem::matmulA_impl(const double* x, double* y, bool sx_ready, bool finish) {
    CMDR_REQUIRE(finalized_, "finalize first");
    const int ncomp = (int)comps_.size();
    span_begin(3);
    // sqrtS_x = S^1/2 x  (comm_cr_mod.f90:792-836)
    if (!sx_ready)
        launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 0, x, nullptr, sx_.get(), false, stream_);
    for (Compact& K : compacts_) {                                                   // pamp * P_cg(2)  :817-833
        if (K.active) launch_vec_scale(0, x + K.pos, K.sigma_dev.get(), nullptr, nullptr, sx_.get() + K.pos, K.nparam, stream_);
        CMDR_HIP_CHECK(hipMemsetAsync(yc_.get() + K.pos, 0, sizeof(double) * K.nparam, stream_));
    }
    for (Group& G : groups_) {   // per-band loop :843-954, all bands of a geometry batched
        ShtPlan& P = *G.plan;
        const double* extra = nullptr;
        if (!G.mix.empty()) { mix_forward(G, sx_.get()); extra = G.E.get(); }            // varying mixing :2082-2084
        const double* wfwd = literal_quirks_ ? G.w_fwd.get() : G.w.get();
        if (pipelined(G))   // the batches of the pipelined form read the stream
            launch_band_prep(comps_dev_.get(), ncomp, sx_.get(), wfwd, G.bm_stokes_dev.get(), P.stream(),
                             P.leg().cnorm.get(), G.lmax, G.nT, stream_, extra);
        if (G.npol)
            launch_band_prep2(comps_dev_.get(), ncomp, sx_.get(), wfwd, G.nT, P.stream2(), G.npol,
                              P.leg2().cnorm.get(), G.lmax, stream_, extra);
        if (group_has_compact(G)) {
            // compact objects live in pixel space (:872-897, :935-948): the map has to exist, so this plan runs the ring
            // stage unfused: phases -> map, + P a, * N^-1, P^t ., map -> phases
            const int64_t np = P.npix_local();
            G.tmpmap.ensure((size_t)G.nbm * np);
            span_begin(0);
            synth_T_of(G, sx_.get(), wfwd, extra);
            if (G.npol) P.synth2_from_stream(G.npol, G.nT, stream_);
            span_end();
            span_begin(1);
            P.rings(0, G.tmpmap.get(), np, nullptr, false, G.nbm, stream_);
            compact_forward(G, sx_.get(), G.tmpmap.get());
            for (int bm = 0; bm < G.nbm; ++bm) {
                const Band& B = bands_[G.bm_band[bm]];
                if (B.qucov_iN.size()) {
                    if (G.bm_stokes[bm] == 0) qucov_invN(G, G.bm_band[bm], G.tmpmap.get());
                    continue;
                }
                launch_pix(0, B.mul.get() + (int64_t)G.bm_stokes[bm] * np, G.tmpmap.get() + (int64_t)bm * np, nullptr,
                           G.tmpmap.get() + (int64_t)bm * np, np, stream_);
            }
            compact_adjoint(G, G.tmpmap.get(), yc_.get());
            P.rings(1, G.tmpmap.get(), np, nullptr, false, G.nbm, stream_);
            span_end();
            continue;
        }
        if (pipelined(G)) {
            // batches of 3 maps: Y of batch j+1 (main stream, VALU-bound) beside N^-1 of batch j (ring stream, LDS-bound)
            const int nbatch = (G.nT + kPipeBatch - 1) / kPipeBatch;
            pipeline_events(G, nbatch);
            span_begin(0);
            for (int j = 0; j < nbatch; ++j) {
                const int k0 = j * kPipeBatch, nb = std::min(kPipeBatch, G.nT - k0);
                P.synth_range(k0, nb, G.nT, stream_);
                CMDR_HIP_CHECK(hipEventRecord(G.ev_synth[j], stream_));
                CMDR_HIP_CHECK(hipStreamWaitEvent(stream_ring_, G.ev_synth[j], 0));
                span_begin(1, stream_ring_);
                P.rings_fused_range(k0, nb, G.mul_ptrs.get(), stream_ring_);
                span_end();
                CMDR_HIP_CHECK(hipEventRecord(G.ev_ring[j], stream_ring_));
            }
            span_end();
            G.ring_pending = true;
            continue;
        }
        span_begin(0);
        synth_T_of(G, sx_.get(), wfwd, extra);                                       // Y        :891 (T: spin 0)
        if (G.npol) { span_begin(6); P.synth2_from_stream(G.npol, G.nT, stream_); span_end(); }   // (Q,U): spin 2, comm_map_mod.f90:446
        span_end();
        span_begin(1);
        P.rings(2, nullptr, 0, G.mul_ptrs.get(), false, G.nbm, stream_, G.that.get());   // N^-1 :905 fused with both FFTs
        span_end();
    }
    adjoint_groups_to_yc(false);                                                     // Yt :915, projectBand :920-948
    if (!finish) { span_end(); return; }
    // y = S^1/2 yc + x  (:957-1008)
    launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 0, yc_.get(), x, y, false, stream_);
    for (Compact& K : compacts_) {                                                   // :985-1003
        if (K.active) launch_vec_scale(2, yc_.get() + K.pos, K.sigma_dev.get(), x + K.pos, nullptr, y + K.pos, K.nparam, stream_);
        else CMDR_HIP_CHECK(hipMemsetAsync(y + K.pos, 0, sizeof(double) * K.nparam, stream_));
    }
    span_end();
}

// compute_residual(band, cg_samp_group) (comm_chisq_mod.f90:196-267): resid_b = data_b - Y sum_{c not in the group}
// getBand_c(alm) - sum_{compact c not in the group} getBand_c, from the components' own amplitudes in stacked layout
// (cr_amp2x order, physical units: no S^-1/2).  The forward half of the matvec with the active flags inverted.
void CrSystem::flip_active() {
    for (Comp& C : comps_) C.d.active = C.d.active ? 0 : 1;
    for (Compact& K : compacts_) K.active = K.active ? 0 : 1;
    std::vector<CompDev> cd;
    for (Comp& C : comps_) cd.push_back(C.d);
    comps_dev_.upload(cd);
    rebuild_weights();
    rebuild_mixing();
}

// band signal maps of the components whose active flag is set, into G.tmpmap [nbm][npix_local]:
// T maps of a plan: weighted component sum -> Legendre synthesis.  In the workgroup form the synthesis kernel forms
// the coefficients while it stages its tiles (no stream written); otherwise k_band_prep writes the stream first.
void CrSystem::synth_T_of(Group& G, const double* v, const double* w, const double* extra) {
    ShtPlan& P = *G.plan;
    const int ncomp = (int)comps_.size();
    if (P.can_prep() && ncomp <= 8) {   // the kernel's term table holds 5 maps x 8 components
        const PrepDev prep{comps_dev_.get(), ncomp, v, w, G.bm_stokes_dev.get(), P.leg().cnorm.get(), extra};
        P.synth_from_prep(prep, G.nT, stream_);
        return;
    }
    launch_band_prep(comps_dev_.get(), ncomp, v, w, G.bm_stokes_dev.get(), P.stream(), P.leg().cnorm.get(), G.lmax, G.nT,
                     stream_, extra);
    P.synth_from_stream(G.nT, stream_);
}

// Y sum_c getBand_c(alm) + pixel-space getBand of the compact ones; sx = amplitudes as they enter getBand
void CrSystem::forward_maps(Group& G, const double* sx) {
    const int ncomp = (int)comps_.size();
    ShtPlan& P = *G.plan;
    const int64_t np = P.npix_local();
    const double* extra = nullptr;
    if (!G.mix.empty()) { mix_forward(G, sx); extra = G.E.get(); }
    if (G.npol)
        launch_band_prep2(comps_dev_.get(), ncomp, sx, G.w.get(), G.nT, P.stream2(), G.npol, P.leg2().cnorm.get(), G.lmax,
                          stream_, extra);
    G.tmpmap.ensure((size_t)G.nbm * np);
    synth_T_of(G, sx, G.w.get(), extra);
    if (G.npol) P.synth2_from_stream(G.npol, G.nT, stream_);
    P.rings(0, G.tmpmap.get(), np, nullptr, false, G.nbm, stream_);
    compact_forward(G, sx, G.tmpmap.get());
}

void CrSystem::compute_residual(const double* amp, const double* const* data, double* const* resid) {
    CMDR_REQUIRE(finalized_, "finalize first");
    sync();
    flip_active();
    for (Group& G : groups_) {
        const int64_t np = G.plan->npix_local();
        forward_maps(G, amp);                                                  // res%Y() + ptsrc%map
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            launch_axpby(data[b] + (int64_t)j * np, G.tmpmap.get() + (int64_t)bm * np, -1.0, resid[b] + (int64_t)j * np, np,
                         stream_);
        }
    }
    sync();
    flip_active();
}

// applyMonoDipolePrior (comm_diffuse_comp_mod.f90:5738-5827).  The reference copies the component (comm_map(self%x)),
// convolves with the output beam, synthesises the map (map%Y; only column 1 is used), fits a monopole (mask-weighted
// mean, :5764-5768) or monopole + dipole (4 x 4 normal equations over the pixels with mask >= 0.5, :5779-5794, solved with
// dgesv) and subtracts the fit from four a_lm entries (:5811-5824) -- and from self%x%map (:5771, :5797-5801), a pixel
// buffer of the driver that the CR path never reads (not mirrored here).  Here: beam (k_alm_copy) -> one Legendre
// synthesis + ring transform on the component's rings -> per-ring-pair sums (k_md_sums, fixed order) -> sum over ranks
// -> host 4 x 4 LU with partial pivoting -> four element edits.
static void solve4_lu(double A[4][4], double* b, double* x) {
    int piv[4] = {0, 1, 2, 3};
    for (int k = 0; k < 4; ++k) {
        int p = k;
        for (int i = k + 1; i < 4; ++i) if (std::fabs(A[piv[i]][k]) > std::fabs(A[piv[p]][k])) p = i;
        std::swap(piv[k], piv[p]);
        const double d = A[piv[k]][k];
        CMDR_REQUIRE(d != 0.0, "applyMonoDipolePrior: singular normal equations (mask leaves too few pixels)");
        for (int i = k + 1; i < 4; ++i) {
            const double f = A[piv[i]][k] / d;
            for (int j = k; j < 4; ++j) A[piv[i]][j] -= f * A[piv[k]][j];
            b[piv[i]] -= f * b[piv[k]];
        }
    }
    for (int k = 3; k >= 0; --k) {
        double s = b[piv[k]];
        for (int j = k + 1; j < 4; ++j) s -= A[piv[k]][j] * x[j];
        x[k] = s / A[piv[k]][k];
    }
}

void CrSystem::apply_mono_dipole_prior(int comp, double* amp, int nside, const double* b_l_out, const double* mask,
                                       int type, double* mu) {
    CMDR_REQUIRE(finalized_, "finalize first");
    CMDR_REQUIRE(comp >= 0 && comp < (int)comps_.size(), "bad component index");
    CMDR_REQUIRE(type == 1 || type == 2, "mono prior type must be 1 (monopole) or 2 (monopole+dipole)");
    CMDR_REQUIRE(amp && mask && mu && nside >= 1, "bad arguments");
    const CompDev& C = comps_[comp].d;
    const int lmax = C.lmax;
    // the component's own map geometry: a band plan of the same (nside, lmax) if there is one, else a plan of its own
    ShtPlan* P = nullptr;
    for (Group& G : groups_)
        if (G.nside == nside && G.lmax == lmax) { P = G.plan.get(); break; }
    if (!P) {
        const int64_t key = (int64_t)nside * 65536 + lmax;
        if (!md_plans_.count(key)) {
            std::vector<int> rings;
            for (auto& rs : ring_sets_) if (rs.first == nside) rings = rs.second;
            md_plans_[key] = std::make_unique<ShtPlan>(nside, lmax, rings, nullptr, 1);
        }
        P = md_plans_[key].get();
    }
    const int64_t na = nalm_packed(lmax), np = P->npix_local();
    const int npair = P->npair();
    md_alm_.ensure((size_t)na);
    md_map_.ensure((size_t)std::max<int64_t>(np, 1));
    md_part_.ensure((size_t)std::max(npair, 1) * kMdSums + kMdSums);
    const double* bl = nullptr;
    if (b_l_out) {
        md_bl_.upload(std::vector<double>(b_l_out, b_l_out + lmax + 1));
        bl = md_bl_.get();
    }
    launch_alm_copy(amp + C.pos, lmax, md_alm_.get(), lmax, bl, false, stream_, lmax);      // B_out%conv, T column
    P->alm2map(md_alm_.get(), na, md_map_.get(), np, 1, false, stream_);                    // map%Y
    P->md_sums(md_map_.get(), mask, type, md_part_.get(), stream_);
    sync();
    std::vector<double> part((size_t)npair * kMdSums);
    if (npair) CMDR_HIP_CHECK(hipMemcpy(part.data(), md_part_.get(), sizeof(double) * part.size(), hipMemcpyDeviceToHost));
    double S[kMdSums];
    for (int k = 0; k < kMdSums; ++k) S[k] = 0.0;
    for (int p = 0; p < npair; ++p)
        for (int k = 0; k < kMdSums; ++k) S[k] += part[(size_t)p * kMdSums + k];
    if (allreduce_ || allreduce_s_ || rccl_.ready()) {   // the mpi_allreduce of a, b / Amat, bmat over the rings' owners
        double* dv = md_part_.get() + (size_t)npair * kMdSums;
        CMDR_HIP_CHECK(hipMemcpy(dv, S, sizeof(S), hipMemcpyHostToDevice));
        reduce(dv, kMdSums);
        sync();
        CMDR_HIP_CHECK(hipMemcpy(S, dv, sizeof(S), hipMemcpyDeviceToHost));
    }
    mu[0] = mu[1] = mu[2] = mu[3] = 0.0;
    if (type == 1) {
        CMDR_REQUIRE(S[1] != 0.0, "applyMonoDipolePrior: the mask is empty");
        mu[0] = S[0] / S[1];
    } else {
        double A[4][4] = {{S[0], S[1], S[2], S[3]}, {S[1], S[4], S[5], S[6]}, {S[2], S[5], S[7], S[8]}, {S[3], S[6], S[8], S[9]}};
        double b[4] = {S[10], S[11], S[12], S[13]};
        solve4_lu(A, b, mu);
    }
    // (0,0) -= mu0 sqrt(4 pi); (1,-1) -= mu2 sqrt(4 pi / 3); (1,0) -= mu3 sqrt(4 pi / 3); (1,1) += mu1 sqrt(4 pi / 3)
    const double pi = 3.14159265358979323846, s0 = std::sqrt(4.0 * pi), s1 = std::sqrt(4.0 * pi / 3.0);
    double e[4];
    const int64_t i00 = d_packed_index(lmax, 0, 0), i10 = d_packed_index(lmax, 1, 0), i11 = d_packed_index(lmax, 1, 1);
    CMDR_HIP_CHECK(hipMemcpy(&e[0], amp + C.pos + i00, sizeof(double), hipMemcpyDeviceToHost));
    e[0] -= mu[0] * s0;
    CMDR_HIP_CHECK(hipMemcpy(amp + C.pos + i00, &e[0], sizeof(double), hipMemcpyHostToDevice));
    if (lmax >= 1) {
        CMDR_HIP_CHECK(hipMemcpy(&e[1], amp + C.pos + i10, sizeof(double), hipMemcpyDeviceToHost));
        CMDR_HIP_CHECK(hipMemcpy(&e[2], amp + C.pos + i11, 2 * sizeof(double), hipMemcpyDeviceToHost));   // (+1, -1)
        e[1] -= mu[3] * s1;
        e[2] += mu[1] * s1;
        e[3] -= mu[2] * s1;
        CMDR_HIP_CHECK(hipMemcpy(amp + C.pos + i10, &e[1], sizeof(double), hipMemcpyHostToDevice));
        CMDR_HIP_CHECK(hipMemcpy(amp + C.pos + i11, &e[2], 2 * sizeof(double), hipMemcpyHostToDevice));
    }
}

// cr_compute_chisq (comm_cr_mod.f90:408-465) -> compute_chisq(chisq_fullsky) (comm_chisq_mod.f90:32-118): with the
// group's amplitudes set to S^1/2 x, chisq = sum_bands sum_pix (sqrtInvN (d - all signal))^2 = || siN (resid - signal
// of the group) ||^2, resid = the maps the last cmdr_compute_rhs received.  sqrtInvN here carries no samp-group mask.
double CrSystem::chisq_of(const double* x) {
    CMDR_REQUIRE(!last_resid_.empty(), "the chisq criterion needs the residual maps: call cmdr_compute_rhs first");
    CMDR_REQUIRE(groups_.size() <= 8, "too many plans for the chisq criterion");
    const int ncomp = (int)comps_.size();
    launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 0, x, nullptr, sx_.get(), false, stream_);
    for (Compact& K : compacts_)
        if (K.active) launch_vec_scale(0, x + K.pos, K.sigma_dev.get(), nullptr, nullptr, sx_.get() + K.pos, K.nparam, stream_);
    double* scal = scal_.get();
    int g = 0;
    for (Group& G : groups_) {
        const int64_t np = G.plan->npix_local();
        forward_maps(G, sx_.get());
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            const Band& B = bands_[b];
            CMDR_REQUIRE(!B.qucov_iN.size(), "the chisq criterion is not available with QU-covariance bands");
            double* t = G.tmpmap.get() + (int64_t)bm * np;
            const double* sraw = (B.siN_raw.size() ? B.siN_raw.get() : B.siN.get()) + (int64_t)j * np;
            launch_axpby(last_resid_[b] + (int64_t)j * np, t, -1.0, t, np, stream_);
            launch_pix(0, sraw, t, nullptr, t, np, stream_);
        }
        launch_dot(G.tmpmap.get(), G.tmpmap.get(), (int64_t)G.nbm * np, dot_partial_.get(), scal, 4 + g, false, stream_);
        ++g;
    }
    sync();
    double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CMDR_HIP_CHECK(hipMemcpy(h, scal + 4, sizeof(double) * g, hipMemcpyDeviceToHost));
    double tot = 0.0;
    for (int k = 0; k < g; ++k) tot += h[k];
    if (allreduce_ || allreduce_s_ || rccl_.ready()) {   // every (band, pixel) lives on exactly one rank in all sharding layouts
        CMDR_HIP_CHECK(hipMemcpy(scal + 4, &tot, sizeof(double), hipMemcpyHostToDevice));
        reduce(scal + 4, 1);
        sync();
        CMDR_HIP_CHECK(hipMemcpy(&tot, scal + 4, sizeof(double), hipMemcpyDeviceToHost));
    }
    return tot;
}

// ------------------------------------------------------------------------------------------------- RHS
void CrSystem::compute_rhs(bool sample, const double* const* resid, const double* const* xi, const double* eta,
                           const double* mu, double* rhs) {
    CMDR_REQUIRE(finalized_, "finalize first");
    const int ncomp = (int)comps_.size();
    // the 'chisq' convergence criterion evaluates against these; a solve that asks for it copies them on entry
    last_resid_.assign(resid, resid + bands_.size());
    resid_owned_ = false;
    for (Compact& K : compacts_) CMDR_HIP_CHECK(hipMemsetAsync(yc_.get() + K.pos, 0, sizeof(double) * K.nparam, stream_));
    for (Group& G : groups_) {
        ShtPlan& P = *G.plan;
        const int64_t np = P.npix_local();
        G.tmpmap.ensure((size_t)G.nbm * np);
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            const Band& B = bands_[b];
            const double* dmap = resid[b] + (int64_t)j * np;
            double* out = G.tmpmap.get() + (int64_t)bm * np;
            if (B.qucov_iN.size()) {   // dense noise: T = 0, (Q;U) through the matrices (comm_N_QUcov_mod.f90:320-385)
                if (j != 0) continue;
                const CellBase cb = cell_base(b);
                const int n2 = (int)(2 * np);
                CMDR_HIP_CHECK(hipMemsetAsync(G.tmpmap.get() + cb.off[0], 0, sizeof(double) * np, stream_));
                if (sample) {
                    qucov_tmp_.ensure((size_t)n2);
                    qucov_tmp2_.ensure((size_t)n2);
                    launch_dense_mv(B.qucov_siN.get(), resid[b] + np, qucov_tmp_.get(), n2, stream_);
                    launch_axpby(qucov_tmp_.get(), xi[b] + np, 1.0, qucov_tmp2_.get(), n2, stream_);
                    launch_dense_mv(B.qucov_siN.get(), qucov_tmp2_.get(), G.tmpmap.get() + cb.off[1], n2, stream_);
                } else {
                    launch_dense_mv(B.qucov_iN.get(), resid[b] + np, G.tmpmap.get() + cb.off[1], n2, stream_);
                }
                continue;
            }
            if (sample)  // sqrtInvN, + xi, sqrtInvN  (comm_cr_mod.f90:600-609)
                launch_pix(1, B.siN.get() + (int64_t)j * np, dmap, xi[b] + (int64_t)j * np, out, np, stream_);
            else         // invN (:611)
                launch_pix(0, B.mul.get() + (int64_t)j * np, dmap, nullptr, out, np, stream_);
        }
        compact_adjoint(G, G.tmpmap.get(), yc_.get());                               // projectBand of compact objects :661-680
        P.rings(1, G.tmpmap.get(), np, nullptr, false, G.nbm, stream_);              // Yt :615
    }
    adjoint_groups_to_yc(true);                                                      // beam, F_mean :616-639
    // rhs = S^1/2 yc + eta + S^-1/2 mu   (:652-659, :690-728)
    if (sample && only_pol_) {   // no temperature fluctuation term: "if (j == 1 .and. only_pol) cycle" (:705)
        CMDR_HIP_CHECK(hipMemcpyAsync(q_.get(), eta, ncr_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
        for (const Comp& C : comps_)
            CMDR_HIP_CHECK(hipMemsetAsync(q_.get() + C.d.pos, 0, C.d.nalm * sizeof(double), stream_));
        eta = q_.get();
    }
    const double* add = nullptr;
    if (mu) {
        launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 1, mu, sample ? eta : nullptr, tmp_.get(), false,
                     stream_);
        add = tmp_.get();
    } else if (sample) {
        add = eta;
    }
    launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 0, yc_.get(), add, rhs, false, stream_);
    for (Compact& K : compacts_) {   // sigma * P^t(...) + eta + P(1)/P(2)   (:676-679, :750-761)
        if (K.active)
            launch_vec_scale(3, yc_.get() + K.pos, K.sigma_dev.get(), sample ? eta + K.pos : nullptr, K.mean_dev.get(),
                             rhs + K.pos, K.nparam, stream_);
        else CMDR_HIP_CHECK(hipMemsetAsync(rhs + K.pos, 0, sizeof(double) * K.nparam, stream_));
    }
}

// ------------------------------------------------------------------------------------------------- preconditioner
namespace {
// invert_matrix_with_mask (math_tools.f90:406-456) for a small dense block, Gauss-Jordan with partial pivoting
void invert_with_mask(std::vector<double>& A, int n) {
    std::vector<char> mask(n, 1);
    for (int i = 0; i < n; ++i)
        if (std::fabs(A[i * n + i]) <= 0.0) { mask[i] = 0; A[i * n + i] = 1.0; }
    std::vector<double> I(n * n, 0.0);
    for (int i = 0; i < n; ++i) I[i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
        if (A[piv * n + c] == 0.0) throw Error("singular preconditioner block");
        if (piv != c) for (int k = 0; k < n; ++k) { std::swap(A[c * n + k], A[piv * n + k]); std::swap(I[c * n + k], I[piv * n + k]); }
        const double inv = 1.0 / A[c * n + c];
        for (int k = 0; k < n; ++k) { A[c * n + k] *= inv; I[c * n + k] *= inv; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[r * n + c];
            if (f == 0.0) continue;
            for (int k = 0; k < n; ++k) { A[r * n + k] -= f * A[c * n + k]; I[r * n + k] -= f * I[c * n + k]; }
        }
    }
    A = I;
    for (int i = 0; i < n; ++i) if (!mask[i]) A[i * n + i] = 0.0;
}
}  // namespace

void CrSystem::precond_init_diag() {
    CMDR_REQUIRE(finalized_, "finalize first");
    const int nband = (int)bands_.size(), npre = (int)comps_.size();
    // ---- invN_diag per band: compute_invN_lm (comm_N_mod.f90:127-197) as an exact Gauss-Legendre quadrature of
    //      Npix/4pi * Int |Y_lm|^2 g dOmega, g = sum_{l'<=lmax} a_l'0 Y_l'0, a = YtW(siN^2)  (:134)
    std::map<int, std::unique_ptr<LegendreDev>> glplans;
    std::map<int, std::pair<std::vector<double>, std::vector<double>>> glnodes;
    for (int b = 0; b < nband; ++b) {
        Band& B = bands_[b];
        Group& G = groups_[B.group];
        ShtPlan& P = *G.plan;
        const int lmax = B.lmax;
        const int64_t np = P.npix_local(), na = nalm_packed(lmax);
        const int ng = (3 * lmax) / 2 + 2, nhalf = (ng + 1) / 2;
        if (!glplans.count(lmax)) {
            std::vector<double> gx, gw;
            gauss_legendre(ng, gx, gw);
            std::vector<double> x(nhalf), sth(nhalf);
            for (int k = 0; k < nhalf; ++k) { x[k] = std::max(gx[k], 0.0); sth[k] = std::sqrt((1.0 - x[k]) * (1.0 + x[k])); }
            LegendreTables T;
            const int Rg = nhalf >= 1024 ? 4 : (nhalf >= 256 ? 2 : 1);
            T.build(lmax, x, sth, Rg, Rg);
            auto L = std::make_unique<LegendreDev>();
            L->upload(T);
            glplans[lmax] = std::move(L);
            glnodes[lmax] = {gx, gw};
        }
        LegendreDev& L = *glplans[lmax];
        const auto& gx = glnodes[lmax].first;
        const auto& gw = glnodes[lmax].second;
        DevBuf<double> siN2(np), alm(na), ph((size_t)L.ph_elems()), part((size_t)L.nchunk * L.tri_elems());
        DevBuf<double> wn(L.npair_pad), ws(L.npair_pad);
        part.zero(stream_);
        B.invN_diag.alloc((size_t)na * B.nmaps);
        B.invN_diag_h.assign((size_t)na * B.nmaps, 0.0);
        for (int j = 0; j < B.nmaps; ++j) {
            const double* sraw = (B.siN_raw.size() ? B.siN_raw.get() : B.siN.get()) + (int64_t)j * np;
            launch_pix(0, sraw, sraw, nullptr, siN2.get(), np, stream_);   // invN_diag%map = siN**2 (comm_N_rms_mod.f90:217)
            P.map2alm(siN2.get(), np, alm.get(), na, 1, true, stream_);
            reduce_rings(alm.get(), na);
            std::vector<double> al0(lmax + 1);
            sync();
            CMDR_HIP_CHECK(hipMemcpy(al0.data(), alm.get(), sizeof(double) * (lmax + 1), hipMemcpyDeviceToHost));
            // g(theta_k) and node weights
            std::vector<double> g(ng);
            for (int k = 0; k < ng; ++k) {
                const double x = gx[k];
                double lp = 0.0, lc = std::sqrt(1.0 / (4.0 * kPi)), sacc = 0.0;
                for (int l = 0;; ++l) {
                    sacc += al0[l] * lc;
                    if (l == lmax) break;
                    const double dl = l, dl1 = l + 1;
                    const double e0 = std::sqrt(dl * dl / (4.0 * dl * dl - 1.0));
                    const double e1 = std::sqrt(dl1 * dl1 / (4.0 * dl1 * dl1 - 1.0));
                    const double ln = (x * lc - e0 * lp) / e1;
                    lp = lc;
                    lc = ln;
                }
                g[k] = sacc;
            }
            const double npix_full = 12.0 * B.nside * (double)B.nside;
            std::vector<double> hwn(L.npair_pad, 0.0), hws(L.npair_pad, 0.0);
            for (int k = 0; k < nhalf; ++k) {
                const int ks = ng - 1 - k;
                const double f = 2.0 * kPi * npix_full / (4.0 * kPi);
                hwn[k] = gw[k] * g[k] * f;
                hws[k] = (ks != k) ? gw[ks] * g[ks] * f : 0.0;
            }
            wn.upload(hwn, stream_);
            ws.upload(hws, stream_);
            launch_fill_gl(ph.get(), wn.get(), ws.get(), L.npair_pad, lmax, stream_);
            launch_leg_adj(L.args(), L.tasks.get(), L.ntasks, ph.get(), L.ph_elems(), part.get(),
                           (int64_t)L.nchunk * L.tri_elems(), L.tri_elems(), 1, true, stream_);
            launch_part_to_diag(part.get(), L.tri_elems(), L.nchunk, L.cnorm.get(), B.invN_diag.get() + (int64_t)j * na,
                                lmax, stream_);
        }
        sync();
        CMDR_HIP_CHECK(hipMemcpy(B.invN_diag_h.data(), B.invN_diag.get(), sizeof(double) * na * B.nmaps,
                                 hipMemcpyDeviceToHost));
    }
    // ---- M0 = sum_bands invN_diag * b_l^2 * F F   (initDiffPrecond_diagonal, comm_diffuse_comp_mod.f90:1199-1230)
    lmax_pre_ = -1;
    nmaps_pre_ = 0;
    for (const Comp& C : comps_) { lmax_pre_ = std::max(lmax_pre_, C.d.lmax); nmaps_pre_ = std::max(nmaps_pre_, C.d.nmaps); }
    const int64_t nt = ntri(lmax_pre_);
    M0_.assign((size_t)nmaps_pre_ * npre * npre * nt, 0.0);
    for (int j = 0; j < nmaps_pre_; ++j)
        for (int q = 0; q < nband; ++q) {
            const Band& B = bands_[q];
            if (j >= B.nmaps) continue;
            const int64_t na = nalm_packed(B.lmax);
            for (int m = 0; m <= std::min(lmax_pre_, B.lmax); ++m)
                for (int l = m; l <= std::min(lmax_pre_, B.lmax); ++l) {
                    const int64_t t = moff(lmax_pre_, m) + (l - m);
                    const int64_t i2 = mind(B.lmax, m) + (m == 0 ? l : 2 * (l - m));
                    const double bl = B.b_l[l + (size_t)(B.lmax + 1) * j];
                    const double base = B.invN_diag_h[i2 + (size_t)na * j] * bl * bl;
                    for (int k1 = 0; k1 < npre; ++k1) {
                        const Comp& p1 = comps_[k1];
                        if (l > p1.d.lmax || j >= p1.d.nmaps) continue;
                        for (int k2 = 0; k2 < npre; ++k2) {
                            const Comp& p2 = comps_[k2];
                            if (l > p2.d.lmax || j >= p2.d.nmaps) continue;
                            M0_[(((size_t)j * npre + k1) * npre + k2) * nt + t] +=
                                base * p1.F_mean[q + (size_t)nband * j] * p2.F_mean[q + (size_t)nband * j];
                        }
                    }
                }
        }
    compact_precond_init();
    if (band_sharded_) {   // M0 is a sum over bands: complete it over the ranks; every ring group contributes its bands
                           // ring_replicas_ times
        DevBuf<double> tmp(M0_.size());
        tmp.upload(M0_, stream_);
        reduce(tmp.get(), (int64_t)M0_.size());
        sync();
        CMDR_HIP_CHECK(hipMemcpy(M0_.data(), tmp.get(), sizeof(double) * M0_.size(), hipMemcpyDeviceToHost));
        const double f = 1.0 / (double)ring_replicas_;
        for (double& v : M0_) v *= f;
    }
    precond_ready_ = false;
}

// ------------------------------------------------------------------------------------------------- low-l preconditioner
void CrSystem::set_lowl(int comp, int L, const int* nside_lowres, const double* const* siN_lowres) {
    CMDR_REQUIRE(finalized_, "finalize first");
    CMDR_REQUIRE(comp >= 0 && comp < (int)comps_.size(), "bad comp");
    sync();
    for (size_t i = 0; i < lowl_.size(); ++i)
        if (lowl_[i].comp == comp) { lowl_.erase(lowl_.begin() + (long)i); break; }
    if (L < 0) return;
    CMDR_REQUIRE(L <= comps_[comp].d.lmax, "lmax_pre_lowl exceeds the component's lmax");
    CMDR_REQUIRE(L <= 200, "lmax_pre_lowl too large for a dense block");
    CMDR_REQUIRE(nside_lowres && siN_lowres, "low-resolution noise maps are NULL");
    lowl_.emplace_back();
    LowL& W = lowl_.back();
    W.comp = comp;
    W.L = L;
    for (int b = 0; b < (int)bands_.size(); ++b) {
        const int ns = nside_lowres[b];
        CMDR_REQUIRE(ns >= 1 && (ns & (ns - 1)) == 0 && siN_lowres[b], "bad low-resolution noise map");
        const int64_t np = 12 * (int64_t)ns * ns;
        W.nside.push_back(ns);
        W.iN.emplace_back((size_t)np);
        for (int64_t i = 0; i < np; ++i) W.iN.back()[i] = siN_lowres[b][i] * siN_lowres[b][i];   // InvN_lowres, comm_N_rms_mod.f90:281
        W.iN_dev.push_back(std::make_unique<DevBuf<double>>());
        W.iN_dev.back()->upload(W.iN.back());
    }
    const int n = (L + 1) * (L + 1);
    const CompDev& C = comps_[comp].d;
    std::vector<int64_t> idx(n);
    for (int l = 0; l <= L; ++l)
        for (int m = -l; m <= l; ++m) {
            const int am = m < 0 ? -m : m;
            idx[l * l + l + m] = C.pos + mind(C.lmax, am) + (am == 0 ? l : 2 * (l - am) + (m < 0 ? 1 : 0));
        }
    W.idx.upload(idx);
    W.xl.alloc(n);
    W.yl.alloc(n);
}

// updateLowlPrecond: the dense block delta + S^1/2 [sum_bands F b_l Yt N_low^-1 Y b_l F] S^1/2 on the temperature a_lm with
// l <= L, probed column by column through low-resolution transforms (lmax 2 L), then inverted.
void CrSystem::lowl_update(LowL& W) {
    const Comp& C = comps_[W.comp];
    const int L = W.L, l2 = 2 * L, n = (L + 1) * (L + 1);
    const int64_t na2 = nalm_packed(l2);
    const int nb = (int)bands_.size();
    std::vector<double> M((size_t)n * n, 0.0);
    auto packed2 = [&](int l, int m) {   // position of (l, m) in the packed a_lm of lmax 2 L
        const int am = m < 0 ? -m : m;
        return mind(l2, am) + (am == 0 ? l : 2 * (l - am) + (m < 0 ? 1 : 0));
    };
    const int B = std::min(n, 16);
    std::vector<double> hin((size_t)B * na2), hout((size_t)B * na2);
    DevBuf<double> din((size_t)B * na2), dout((size_t)B * na2);
    for (int b = 0; b < nb; ++b) {
        const Band& Bd = bands_[b];
        const int ns = W.nside[b];
        const int key = ns * 65536 + l2;
        if (!lowl_plans_.count(key)) lowl_plans_[key] = std::make_unique<ShtPlan>(ns, l2, std::vector<int>{}, nullptr, 16);
        ShtPlan& P = *lowl_plans_[key];
        const int64_t np = P.npix_local();
        DevBuf<double> maps((size_t)B * np);
        std::vector<double> wl(L + 1, 0.0);    // sqrt(S)_TT * F_mean * b_l * mb_eff  (:5128, :5139-5141)
        for (int l = 0; l <= L; ++l) {
            double sq = 1.0;
            if (C.d.lmax_cl >= 0) sq = l <= C.d.lmax_cl ? C.sqrtS[(size_t)C.d.nmaps * C.d.nmaps * l] : 0.0;
            const double bl = l <= Bd.lmax ? Bd.b_l[l] * Bd.mb_eff : 0.0;
            wl[l] = sq * C.F_mean[b] * bl;
        }
        for (int j0 = 0; j0 < n; j0 += B) {
            const int nbat = std::min(B, n - j0);
            std::fill(hin.begin(), hin.end(), 0.0);
            for (int k = 0; k < nbat; ++k) {
                const int i = j0 + k, l = (int)std::floor(std::sqrt((double)i)), m = i - l * l - l;
                hin[(size_t)k * na2 + packed2(l, m)] = wl[l];
            }
            CMDR_HIP_CHECK(hipMemcpyAsync(din.get(), hin.data(), sizeof(double) * nbat * na2, hipMemcpyHostToDevice, stream_));
            P.alm2map(din.get(), na2, maps.get(), np, nbat, false, stream_);                          // Y   :5147
            for (int k = 0; k < nbat; ++k)
                launch_pix(0, W.iN_dev[b]->get(), maps.get() + (int64_t)k * np, nullptr, maps.get() + (int64_t)k * np, np, stream_);
            P.map2alm(maps.get(), np, dout.get(), na2, nbat, false, stream_);                         // Yt  :5159
            sync();
            CMDR_HIP_CHECK(hipMemcpy(hout.data(), dout.get(), sizeof(double) * nbat * na2, hipMemcpyDeviceToHost));
            for (int k = 0; k < nbat; ++k)
                for (int lp = 0; lp <= L; ++lp)
                    for (int mp = -lp; mp <= lp; ++mp)
                        M[(size_t)(j0 + k) * n + lp * lp + lp + mp] += wl[lp] * hout[(size_t)k * na2 + packed2(lp, mp)];
        }
    }
    if (band_sharded_) {   // sum over the band groups; every ring replica of a group computed the same (full-sky) terms
        DevBuf<double> tmp(M.size());
        tmp.upload(M, stream_);
        reduce(tmp.get(), (int64_t)M.size());
        sync();
        CMDR_HIP_CHECK(hipMemcpy(M.data(), tmp.get(), sizeof(double) * M.size(), hipMemcpyDeviceToHost));
        for (double& v : M) v /= (double)ring_replicas_;
    }
    for (int i = 0; i < n; ++i) M[(size_t)i * n + i] += 1.0;                                          // :5205
    // symmetric positive definite: Cholesky M = G G^t, inverse = G^-t G^-1 (invert_matrix(cholesky=.true.) :5229)
    for (int j = 0; j < n; ++j) {
        double d = M[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= M[(size_t)j * n + k] * M[(size_t)j * n + k];
        CMDR_REQUIRE(d > 0.0, "low-l preconditioner block is not positive definite");
        d = std::sqrt(d);
        M[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]);
            for (int k = 0; k < j; ++k) v -= M[(size_t)i * n + k] * M[(size_t)j * n + k];
            M[(size_t)i * n + j] = v / d;
        }
    }
    std::vector<double> Gi((size_t)n * n, 0.0);               // G^-1, lower triangular
    for (int c = 0; c < n; ++c) {
        Gi[(size_t)c * n + c] = 1.0 / M[(size_t)c * n + c];
        for (int i = c + 1; i < n; ++i) {
            double v = 0.0;
            for (int k = c; k < i; ++k) v -= M[(size_t)i * n + k] * Gi[(size_t)k * n + c];
            Gi[(size_t)i * n + c] = v / M[(size_t)i * n + i];
        }
    }
    std::vector<double> Minv((size_t)n * n, 0.0);
    host_parallel_for(n, [&](int i) {
        for (int j = 0; j <= i; ++j) {
            double v = 0.0;
            for (int k = i; k < n; ++k) v += Gi[(size_t)k * n + i] * Gi[(size_t)k * n + j];
            Minv[(size_t)i * n + j] = v;
            Minv[(size_t)j * n + i] = v;
        }
    });
    W.Minv.upload(Minv, stream_);
    W.ready = true;
}

void CrSystem::precond_update_diag() {
    CMDR_REQUIRE(!M0_.empty(), "precond_init_diag first");
    const int npre = (int)comps_.size();
    const int64_t nt = ntri(lmax_pre_);
    std::vector<double> P((size_t)nmaps_pre_ * npre * npre * nt, 0.0);
    host_parallel_for(lmax_pre_ + 1, [&](int m) {
        std::vector<double> M;
        std::vector<int> idx;
        for (int j = 0; j < nmaps_pre_; ++j)
            for (int l = m; l <= lmax_pre_; ++l) {
                const int64_t t = moff(lmax_pre_, m) + (l - m);
                auto at = [&](std::vector<double>& V, int k1, int k2) -> double& {
                    return V[(((size_t)j * npre + k1) * npre + k2) * nt + t];
                };
                idx.clear();
                for (int k = 0; k < npre; ++k) if (at(M0_, k, k) > 0.0) idx.push_back(k);   // comp2ind :1232-1239
                for (int k = 0; k < npre; ++k) at(P, k, k) = 1.0;                             // absent: pass through
                const int n = (int)idx.size();
                if (n == 0) continue;
                M.assign((size_t)n * n, 0.0);
                for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) M[a * n + b] = at(M0_, idx[a], idx[b]);
                for (int a = 0; a < n; ++a) {                                                 // S^1/2 (diag) both sides :1352-1424
                    const Comp& C = comps_[idx[a]];
                    if (C.d.lmax_cl < 0) continue;
                    double dsc = 0.0;
                    if (l <= C.d.lmax_cl && j < C.d.nmaps)
                        dsc = std::sqrt(C.S[j + (size_t)C.d.nmaps * (j + (size_t)C.d.nmaps * l)]);
                    for (int b = 0; b < n; ++b) { M[a * n + b] *= dsc; M[b * n + a] *= dsc; }
                }
                if (only_pol_ && j == 0) std::fill(M.begin(), M.end(), 0.0);                  // :1428-1433
                for (int a = 0; a < n; ++a) {                                                 // add unity :1452-1470
                    const Comp& C = comps_[idx[a]];
                    if (C.d.lmax_cl < 0) continue;
                    if (l <= C.d.lmax) M[a * n + a] += 1.0;
                }
                for (int a = 0; a < n; ++a) {                                                 // inactive comps :1484-1495
                    if (comps_[idx[a]].d.active) continue;
                    for (int b = 0; b < n; ++b) { M[a * n + b] = 0.0; M[b * n + a] = 0.0; }
                }
                bool any = false;
                for (double v : M) if (v != 0.0) { any = true; break; }
                if (any) invert_with_mask(M, n);                                              // :1539-1551
                for (int a = 0; a < n; ++a) {
                    at(P, idx[a], idx[a]) = 0.0;
                    for (int b = 0; b < n; ++b) at(P, idx[a], idx[b]) = M[a * n + b];
                }
            }
    });
    P_.upload(P, stream_);
    precond_type_ = 0;
    precond_ready_ = true;
    for (LowL& W : lowl_) lowl_update(W);      // update_precond rebuilds the low-l block every time (comm_cr_mod.f90:1136-1147)
}

// ------------------------------------------------------------------------------------------------- pseudo-inverse
namespace {
// Moore-Penrose pseudo-inverse of a small m x n matrix (row-major, m >= n) by one-sided Jacobi SVD; singular values
// below thr * s_max are dropped (compute_pseudo_inverse, math_tools.f90:234-292: DGESVD, threshold 1e-12).
// out: n x m row-major.
void pseudo_inverse(const std::vector<double>& A, int m, int n, double thr, std::vector<double>& out) {
    std::vector<double> U(A), V((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                double a = 0.0, b = 0.0, c = 0.0;
                for (int i = 0; i < m; ++i) { a += U[i * n + p] * U[i * n + p]; b += U[i * n + q] * U[i * n + q]; c += U[i * n + p] * U[i * n + q]; }
                if (c == 0.0) continue;
                off = std::max(off, std::fabs(c) / std::sqrt(std::max(a * b, 1e-300)));
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < m; ++i) {
                    const double up = U[i * n + p], uq = U[i * n + q];
                    U[i * n + p] = cs * up - sn * uq;
                    U[i * n + q] = sn * up + cs * uq;
                }
                for (int i = 0; i < n; ++i) {
                    const double vp = V[i * n + p], vq = V[i * n + q];
                    V[i * n + p] = cs * vp - sn * vq;
                    V[i * n + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-15) break;
    }
    std::vector<double> sv(n);
    double smax = 0.0;
    for (int j = 0; j < n; ++j) {
        double a = 0.0;
        for (int i = 0; i < m; ++i) a += U[i * n + j] * U[i * n + j];
        sv[j] = std::sqrt(a);
        smax = std::max(smax, sv[j]);
    }
    out.assign((size_t)n * m, 0.0);
    for (int j = 0; j < n; ++j) {
        if (!(sv[j] > thr * smax) || sv[j] == 0.0) continue;
        const double inv2 = 1.0 / (sv[j] * sv[j]);          // V (1/s) (U_j / s)^T
        for (int r = 0; r < n; ++r)
            for (int i = 0; i < m; ++i) out[(size_t)r * m + i] += V[r * n + j] * inv2 * U[i * n + j];
    }
}
}  // namespace

void CrSystem::precond_init_pseudoinv() {
    CMDR_REQUIRE(finalized_, "finalize first");
    CMDR_REQUIRE(!band_sharded_, "the pseudo-inverse preconditioner needs every band on every rank (ring sharding only)");
    CMDR_REQUIRE(compacts_.empty(), "compact components are supported with the diagonal preconditioner type only");
    for (Group& G : groups_) {
        ShtPlan& P = *G.plan;
        const int64_t np = P.npix_local(), na = P.nalm();
        const std::vector<double> pw = P.pixel_weights();
        DevBuf<double> m2((size_t)2 * np), alm((size_t)2 * na), ones((size_t)2 * np), sums(2);
        { std::vector<double> o((size_t)2 * np, 1.0); ones.upload(o); }
        std::vector<const double*> mp(G.nbm);
        for (int b : G.bands) {
            Band& B = bands_[b];
            B.alpha_nu.assign(B.nmaps, 0.0);
            const double* sraw = B.siN_raw.size() ? B.siN_raw.get() : B.siN.get();
            auto alpha_of = [&](int64_t n) {          // sqrt(sum tau^2 / sum tau)  (comm_N_rms_mod.f90:225-246)
                launch_dot(m2.get(), ones.get(), n, dot_partial_.get(), sums.get(), 0, false, stream_);
                launch_dot(m2.get(), m2.get(), n, dot_partial_.get(), sums.get(), 1, false, stream_);
                reduce_rings(sums.get(), 2);
                double h[2];
                sync();
                CMDR_HIP_CHECK(hipMemcpy(h, sums.get(), sizeof(h), hipMemcpyDeviceToHost));
                return h[0] > 0.0 ? std::sqrt(h[1] / h[0]) : 0.0;
            };
            // tau = Y Yt siN^2  (:221-223)
            launch_pix(0, sraw, sraw, nullptr, m2.get(), np, stream_);
            P.map2alm(m2.get(), np, alm.get(), na, 1, false, stream_);
            reduce_rings(alm.get(), na);
            P.alm2map(alm.get(), na, m2.get(), np, 1, false, stream_);
            B.alpha_nu[0] = alpha_of(np);
            if (B.nmaps == 3) {
                launch_pix(0, sraw + np, sraw + np, nullptr, m2.get(), 2 * np, stream_);
                P.map2alm_spin2(m2.get(), m2.get() + np, alm.get(), alm.get() + na, false, stream_);
                reduce_rings(alm.get(), 2 * na);
                P.alm2map_spin2(alm.get(), alm.get() + na, m2.get(), m2.get() + np, false, stream_);
                B.alpha_nu[1] = B.alpha_nu[2] = alpha_of(2 * np);
            }
            // T operator of the preconditioner: WY . N . YtW  ->  pixel multiplier w^2 rms^2
            std::vector<double> mulP(B.Nmap_h.size());
            for (int j = 0; j < B.nmaps; ++j)
                for (int64_t i = 0; i < np; ++i) mulP[(size_t)j * np + i] = B.Nmap_h[(size_t)j * np + i] * pw[i] * pw[i];
            B.mulP.upload(mulP);
        }
        for (int bm = 0; bm < G.nbm; ++bm) mp[bm] = bands_[G.bm_band[bm]].mulP.get() + (int64_t)G.bm_stokes[bm] * np;
        G.mulP_ptrs.upload(mp);
        G.plan->toeplitz_build(mp, G.thatP, stream_);
    }
    lmax_pre_ = -1;
    nmaps_pre_ = 0;
    for (const Comp& C : comps_) { lmax_pre_ = std::max(lmax_pre_, C.d.lmax); nmaps_pre_ = std::max(nmaps_pre_, C.d.nmaps); }
    pinv_init_ = true;
    precond_ready_ = false;
}

void CrSystem::precond_update_pseudoinv() {
    CMDR_REQUIRE(pinv_init_, "precond_init_pseudoinv first");
    const int nb = (int)bands_.size(), npre = (int)comps_.size(), L1 = lmax_pre_ + 1;
    // pinv(U) per (stokes, l): [j][l][npre][nb + npre]
    std::vector<double> PI((size_t)nmaps_pre_ * L1 * npre * (nb + npre), 0.0);
    host_parallel_for(L1, [&](int l) {
        std::vector<double> mat, inv;
        for (int j = 0; j < nmaps_pre_; ++j) {
            mat.assign((size_t)(nb + npre) * npre, 0.0);
            for (int q = 0; q < nb; ++q) {
                const Band& B = bands_[q];
                if (l > B.lmax || j >= B.nmaps) continue;
                for (int k = 0; k < npre; ++k) {
                    const Comp& C = comps_[k];
                    if (l > C.d.lmax || j >= C.d.nmaps || !C.d.active) continue;
                    double v = B.alpha_nu[j] * B.b_l[l + (size_t)(B.lmax + 1) * j] * C.F_mean[q + (size_t)nb * j];
                    if (C.d.lmax_cl >= 0) {                                      // sqrt(getCl(l, j))  :1591-1593
                        double cl = 0.0;
                        if (l <= C.d.lmax_cl)
                            cl = C.cl_diag.empty() ? C.S[j + (size_t)C.d.nmaps * (j + (size_t)C.d.nmaps * l)]
                                                   : C.cl_diag[l + (size_t)(C.d.lmax_cl + 1) * j];
                        v *= std::sqrt(cl);
                    }
                    mat[(size_t)q * npre + k] = v;
                }
            }
            for (int k = 0; k < npre; ++k) {                                     // prior section :1610-1614
                const Comp& C = comps_[k];
                if (C.d.lmax_cl < 0 || l > C.d.lmax || !C.d.active) continue;
                mat[(size_t)(nb + k) * npre + k] = 1.0;
            }
            pseudo_inverse(mat, nb + npre, npre, 1e-12, inv);
            std::copy(inv.begin(), inv.end(), PI.begin() + ((size_t)j * L1 + l) * npre * (nb + npre));
        }
    });
    auto pin = [&](int j, int l, int k, int col) { return PI[(((size_t)j * L1 + l) * npre + k) * (nb + npre) + col]; };
    for (Group& G : groups_) {
        std::vector<double> wi((size_t)G.nbm * npre * (G.lmax + 1), 0.0), wo(wi.size(), 0.0);
        for (int bm = 0; bm < G.nbm; ++bm) {
            const int b = G.bm_band[bm], j = G.bm_stokes[bm];
            const double a2 = bands_[b].alpha_nu[j] * bands_[b].alpha_nu[j];
            for (int k = 0; k < npre; ++k)
                for (int l = 0; l <= std::min(G.lmax, lmax_pre_); ++l) {
                    const double v = j < nmaps_pre_ ? pin(j, l, k, b) : 0.0;
                    wi[((size_t)bm * npre + k) * (G.lmax + 1) + l] = v;
                    wo[((size_t)bm * npre + k) * (G.lmax + 1) + l] = v * a2;   // :2303-2305
                }
        }
        G.w_pin.upload(wi);
        G.w_pout.upload(wo);
    }
    std::vector<double> Q((size_t)nmaps_pre_ * npre * npre * L1, 0.0);             // B B^t, B = prior columns
    for (int j = 0; j < nmaps_pre_; ++j)
        for (int l = 0; l < L1; ++l)
            for (int k1 = 0; k1 < npre; ++k1)
                for (int k2 = 0; k2 < npre; ++k2) {
                    double sacc = 0.0;
                    for (int t = 0; t < npre; ++t) sacc += pin(j, l, k1, nb + t) * pin(j, l, k2, nb + t);
                    Q[(((size_t)j * npre + k1) * npre + k2) * L1 + l] = sacc;
                }
    Qprior_.upload(Q, stream_);
    precond_type_ = 1;
    precond_ready_ = true;
}

// applyDiffPrecond_pseudoinv (comm_diffuse_comp_mod.f90:2238-2380)
void CrSystem::apply_pseudoinv(const double* x, double* y) {
    const int ncomp = (int)comps_.size();
    for (int g = 0; g < (int)groups_.size(); ++g) {
        Group& G = groups_[g];
        ShtPlan& P = *G.plan;
        if (G.npol)
            launch_band_prep2(comps_dev_.get(), ncomp, x, G.w_pin.get(), G.nT, P.stream2(), G.npol,
                              P.leg2().cnorm.get(), G.lmax, stream_);
        synth_T_of(G, x, G.w_pin.get(), nullptr);                                        // (U^+)^t :2279-2291, WY :2295
        if (G.npol) P.synth2_from_stream(G.npol, G.nT, stream_);
        P.rings(2, nullptr, 0, G.mulP_ptrs.get(), false, G.nbm, stream_, G.thatP.get());  // N         :2297
        P.adjoint_to_partials(G.nT, false, stream_);                                     // YtW       :2299
        if (G.npol) P.adjoint2_to_partials(G.npol, G.nT, stream_);
        launch_band_post(comps_dev_.get(), ncomp, lmax_max_, P.partials(), P.part_map_stride(), P.leg().tri_elems(),
                         P.leg().nchunk, G.nT, G.bm_stokes_dev.get(), G.w_pout.get(), P.leg().cnorm.get(), G.lmax,
                         yc_.get(), g > 0, stream_, P.leg().lw_chunk.get());             // alpha^2, U^+  :2303-2322
        if (G.npol)
            launch_band_post2(comps_dev_.get(), ncomp, lmax_max_, P.partials2(), P.part2_pol_stride(), P.leg2().tri4(),
                              P.leg2().nchunk, G.npol, G.w_pout.get(), G.nT, P.leg2().cnorm.get(), G.lmax, yc_.get(),
                              stream_, P.leg2().lw_chunk.get());
    }
    reduce(yc_.get(), ncr_);
    launch_pinv_prior(comps_dev_.get(), ncomp, lmax_max_, Qprior_.get(), lmax_pre_, nmaps_pre_, x, yc_.get(), y,
                      stream_);                                                          // prior terms :2328-2372
}

void CrSystem::invM(const double* x, double* y) {
    CMDR_REQUIRE(precond_ready_, "preconditioner not initialised (precond_init_* + precond_update_*)");
    if (precond_type_ == 1) { apply_pseudoinv(x, y); return; }
    launch_precond_diag(comps_dev_.get(), (int)comps_.size(), P_.get(), lmax_pre_, nmaps_pre_, x, y, stream_);
    for (Compact& K : compacts_)   // applyPtsrcPrecond / applyTemplatePrecond: the block's own dense inverse
        launch_dense_mv(K.Minv.get(), x + K.pos, y + K.pos, K.nparam, stream_);
    for (LowL& W : lowl_) {        // applyLowlPrecond on the INPUT vector's entries (comm_cr_mod.f90:1058-1073)
        if (!W.ready) continue;
        const int n = (W.L + 1) * (W.L + 1);
        launch_index_copy(x, W.idx.get(), W.xl.get(), n, false, stream_);
        launch_dense_mv(W.Minv.get(), W.xl.get(), W.yl.get(), n, stream_);
        launch_index_copy(W.yl.get(), W.idx.get(), y, n, true, stream_);
    }
}

// ------------------------------------------------------------------------------------------------- PCG
SolveResult CrSystem::solve(const double* b, double* x, int crit, double tol, int miniter, int maxiter,
                            int check_freq, const double* x0) {
    CMDR_REQUIRE(finalized_ && precond_ready_, "system / preconditioner not ready");
    CMDR_REQUIRE(check_freq >= 1, "check_freq must be >= 1");
    const bool fixed_iter = (crit == 1), by_chisq = (crit == 2);
    const int ncomp = (int)comps_.size();
    const int64_t n = ncr_;
    if (by_chisq && !resid_owned_) {
        // the criterion reads the residual maps of the last cmdr_compute_rhs at every check: take copies now, so the
        // caller's maps only have to live until this call is entered (not through the solve)
        CMDR_REQUIRE(!last_resid_.empty(), "the chisq criterion needs the residual maps: call cmdr_compute_rhs first");
        resid_own_.resize(bands_.size());
        for (size_t b = 0; b < bands_.size(); ++b) {
            const size_t nb = (size_t)band_npix((int)b) * bands_[b].nmaps;
            resid_own_[b].ensure(nb);
            CMDR_HIP_CHECK(hipMemcpyAsync(resid_own_[b].get(), last_resid_[b], nb * sizeof(double), hipMemcpyDeviceToDevice, stream_));
            last_resid_[b] = resid_own_[b].get();
        }
        resid_owned_ = true;
    }
    double* scal = scal_.get();      // [0] delta_new [1] delta_old [2] d.q [3] delta0
    SolveResult R;
    if (!x0) {                                                                          // :133-134
        CMDR_HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(double), stream_));
        CMDR_HIP_CHECK(hipMemcpyAsync(r_.get(), b, n * sizeof(double), hipMemcpyDeviceToDevice, stream_));  // r = b - A 0
    } else {                                                                            // :136-173
        launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 1, x0, nullptr, x, true, stream_);
        for (Compact& K : compacts_) {                                                  // :160-170
            if (K.active) launch_vec_scale(1, x0 + K.pos, K.sigma_dev.get(), nullptr, nullptr, x + K.pos, K.nparam, stream_);
            else CMDR_HIP_CHECK(hipMemcpyAsync(x + K.pos, x0 + K.pos, sizeof(double) * K.nparam, hipMemcpyDeviceToDevice, stream_));
        }
        matmulA(x, q_.get());
        launch_axpby(b, q_.get(), -1.0, r_.get(), n, stream_);                           // :201
    }
    invM(r_.get(), d_.get());                                                           // :203
    launch_dot(r_.get(), d_.get(), n, dot_partial_.get(), scal, 0, false, stream_);     // :206
    invM(b, tmp_.get());
    launch_dot(b, tmp_.get(), n, dot_partial_.get(), scal, 3, false, stream_);          // :208
    double h[4];
    auto fetch = [&]() {
        sync();
        CMDR_HIP_CHECK(hipMemcpy(h, scal, sizeof(h), hipMemcpyDeviceToHost));
    };
    fetch();
    R.delta0 = h[3];
    const double lim = by_chisq ? tol : tol * h[3];                                     // :220-226
    double chisq = by_chisq ? chisq_of(x) : 0.0;
    // Fused vector updates (3 launches per iteration: k_cg_q, k_cg_xr_precond, k_cg_d_sqrtS) when every entry of the
    // stacked vector is a diffuse a_lm under the diagonal preconditioner; else the general sequence below.
    const char* fuse_env = std::getenv("CMDR_CG_FUSED");                             // read per solve (test hook)
    const bool fuse_on = !fuse_env || std::atoi(fuse_env) != 0;
    bool lowl_on = false;
    for (const LowL& W : lowl_) lowl_on = lowl_on || W.ready;
    const bool fused = fuse_on && precond_type_ == 0 && compacts_.empty() && !lowl_on;
    const int npart = dot_partial_count();
    double* p_dq = nullptr;
    double* p_rs[2] = {nullptr, nullptr};
    bool sx_ready = false;
    if (fused) {
        cg_partials_.ensure((size_t)3 * npart);
        p_dq = cg_partials_.get();
        p_rs[0] = p_dq + npart;
        p_rs[1] = p_dq + 2 * npart;
        launch_cg_seed(scal, 0, p_rs[0], stream_);                                      // delta_new of :206
        launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 0, d_.get(), nullptr, sx_.get(), false, stream_);
        sx_ready = true;
    }
    // m-sliced vectors (set_vector_slicing): this rank's index range of the stacked vector
    const bool sliced = fused && slice_n_ >= 1 && !by_chisq && (rccl_.ready() || allreduce_s_ || allreduce_) &&
                        !band_sharded_ && ncomp == 1 && comps_[0].d.nmaps == 1 && nmaps_pre_ == 1 && groups_.size() == 1 &&
                        groups_[0].npol == 0 && groups_[0].mix.empty() && !pipeline_;
    const int64_t sc = sliced ? slice_count() : 0;
    const int64_t ilo = sliced ? sc * slice_rank_ : 0, ihi = sliced ? std::min<int64_t>(n, sc * (slice_rank_ + 1)) : INT64_MAX;
    if (sliced) {
        const size_t pad = (size_t)(sc * slice_n_);
        if (yc_.size() < pad || sx_.size() < pad || q_.size() < pad) {
            sync();
            // keep what the loop reads from the old buffers: S^1/2 d in sx_ (written just above)
            DevBuf<double> nsx(pad);
            CMDR_HIP_CHECK(hipMemcpy(nsx.get(), sx_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice));
            sx_ = std::move(nsx);
            yc_.alloc(pad);
            q_.alloc(pad);
        }
        CMDR_HIP_CHECK(hipMemsetAsync(yc_.get() + n, 0, sizeof(double) * (pad - n), stream_));
        CMDR_HIP_CHECK(hipMemsetAsync(sx_.get() + n, 0, sizeof(double) * (pad - n), stream_));
    }
    auto fused_iter = [&](int it) {                                                     // one iteration, :253-272
        if (sliced) {
            slice_active_ = true;
            try { matmulA_impl(d_.get(), nullptr, sx_ready, false); } catch (...) { slice_active_ = false; throw; }
            slice_active_ = false;                                                      // yc_[ilo, ihi) = this rank's sums
            launch_cg_q(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), yc_.get(), d_.get(), q_.get(), p_dq, stream_, ilo, ihi);
            reduce(p_dq, npart);                                                        // d.q: block partials over the ranks
            launch_cg_xr_precond(comps_dev_.get(), ncomp, lmax_max_, P_.get(), nmaps_pre_, p_dq, p_rs[(it - 1) & 1],
                                 p_rs[it & 1], x, r_.get(), d_.get(), q_.get(), s_.get(), scal, stream_, ilo, ihi);
            reduce(p_rs[it & 1], npart);                                                // r.s
            launch_cg_d_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), p_rs[(it - 1) & 1], p_rs[it & 1], d_.get(),
                              s_.get(), sx_.get(), scal, stream_, ilo, ihi);
            slice_all_gather(sx_.get());                                                // S^1/2 d for the next synthesis
            sx_ready = true;
            return;
        }
        matmulA_impl(d_.get(), nullptr, sx_ready, false);                               // :253, up to the reduced yc_
        launch_cg_q(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), yc_.get(), d_.get(), q_.get(), p_dq, stream_);
        launch_cg_xr_precond(comps_dev_.get(), ncomp, lmax_max_, P_.get(), nmaps_pre_, p_dq, p_rs[(it - 1) & 1],
                             p_rs[it & 1], x, r_.get(), d_.get(), q_.get(), s_.get(), scal, stream_);   // :254-269
        launch_cg_d_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), p_rs[(it - 1) & 1], p_rs[it & 1], d_.get(),
                          s_.get(), sx_.get(), scal, stream_);                          // :270-272 + head of the next A d
        sx_ready = true;
    };
    int i = 1;
#if !defined(CMDR_EMUL)
    // fixed_iter: no host decision inside the loop, and two consecutive iterations form a fixed launch sequence (the r.s
    // partial buffers alternate by parity) -> iterations 3.. are replays of ONE captured hipGraph of two iterations.  The
    // kernels of small problems (BASELINE configs[1], ring-sharded ranks) are shorter than the host's launch path; the
    // graph removes that path from the loop.  Not with a collective in the matvec (host callbacks cannot be captured,
    // RCCL inside a capture is left for when it can be tested with more than one rank), not while profiling (the event
    // spans are host objects).  CMDR_CG_GRAPH=0 disables; any capture error falls back to eager launches.
    const bool graph_env = [] { const char* e = std::getenv("CMDR_CG_GRAPH"); return !e || std::atoi(e) != 0; }();
    if (graph_env && fused && fixed_iter && !profile_ && !allreduce_ && !allreduce_s_ && !rccl_.ready() && maxiter >= 6) {
        fused_iter(1);                       // eager: lazily sized workspaces, kernel attributes, sx_ready
        fused_iter(2);
        R.niter = 2;
        i = 3;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool captured = false;
        if (hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            bool ok = true;
            try {
                fused_iter(3);
                fused_iter(4);
            } catch (...) {
                ok = false;
            }
            const hipError_t rc = hipStreamEndCapture(stream_, &graph);
            if (ok && rc == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess)
                captured = true;
        }
        (void)hipGetLastError();             // a refused capture leaves a sticky error code behind
        if (captured) {
            for (; i + 1 <= maxiter; i += 2) {
                CMDR_HIP_CHECK(hipGraphLaunch(exec, stream_));
                R.niter = i + 1;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }
#endif
    for (; i <= maxiter; ++i) {                                                         // :230
        if (i % check_freq == 0 && !fixed_iter) {                                       // :236-247
            fetch();
            double val = h[0];
            if (by_chisq) {
                const double prev = chisq;
                chisq = chisq_of(x);                                                    // uses sx_ for S^1/2 x
                sx_ready = false;
                val = std::fabs((prev - chisq) / chisq);
            }
            if (val < lim && (i >= miniter || h[0] <= 1e-30 * h[3])) break;
        }
        if (fused) {
            fused_iter(i);
            R.niter = i;
            continue;
        }
        matmulA(d_.get(), q_.get());                                                    // :253
        launch_dot(d_.get(), q_.get(), n, dot_partial_.get(), scal, 2, false, stream_); // :254
        launch_cg_xr(x, r_.get(), d_.get(), q_.get(), n, scal, 0, 2, stream_);          // :255,:261
        invM(r_.get(), s_.get());                                                       // :266
        launch_dot(r_.get(), s_.get(), n, dot_partial_.get(), scal, 0, true, stream_);  // :269-270
        launch_cg_d(d_.get(), s_.get(), n, scal, 0, 1, stream_);                        // :271-272
        R.niter = i;
    }
    if (sliced) {   // every rank holds its own index range of x: gather the rest (q_ is free now and padded)
        CMDR_HIP_CHECK(hipMemcpyAsync(q_.get() + ilo, x + ilo, sizeof(double) * (ihi - ilo), hipMemcpyDeviceToDevice, stream_));
        if (ihi < sc * (slice_rank_ + 1))
            CMDR_HIP_CHECK(hipMemsetAsync(q_.get() + ihi, 0, sizeof(double) * (sc * (slice_rank_ + 1) - ihi), stream_));
        slice_all_gather(q_.get());
        CMDR_HIP_CHECK(hipMemcpyAsync(x, q_.get(), sizeof(double) * n, hipMemcpyDeviceToDevice, stream_));
    }
    // x <- S^1/2 x  (:350-389)
    launch_sqrtS(comps_dev_.get(), ncomp, lmax_max_, smat_.get(), 0, x, nullptr, tmp_.get(), true, stream_);
    for (Compact& K : compacts_) {                                                      // :376-388
        if (K.active) launch_vec_scale(0, x + K.pos, K.sigma_dev.get(), nullptr, nullptr, tmp_.get() + K.pos, K.nparam, stream_);
        else CMDR_HIP_CHECK(hipMemcpyAsync(tmp_.get() + K.pos, x + K.pos, sizeof(double) * K.nparam, hipMemcpyDeviceToDevice, stream_));
    }
    CMDR_HIP_CHECK(hipMemcpyAsync(x, tmp_.get(), n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    fetch();
    R.delta_new = h[0];
    if (i >= maxiter && !fixed_iter) R.stat = 1;                                        // :392-395
    CMDR_HIP_CHECK(hipGetLastError());
    return R;
}

}  // namespace cmdr
