#include "sht_plan.hpp"

namespace cmdr {

void LegendreDev::upload(const LegendreTables& T) {
    lmax = T.lmax;
    npair_pad = T.npair_pad;
    R = T.R;
    nchunk = T.nchunk;
    Rs = T.Rs;
    synth_wg = T.synth_wg;
    uniform_start = T.uniform_start;
    ntasks_s = (int)T.tasks_s.size();
    tasks_s.upload(T.tasks_s);
    ntasks = (int)T.tasks.size();
    x.upload(T.x);
    ls.upload(T.ls);
    seedc.upload(T.seedc);
    seedp.upload(T.seedp);
    alpha.upload(T.alpha);
    cnorm.upload(T.cnorm);
    tasks.upload(T.tasks);
    lw_chunk.upload(T.lw_chunk);
    tasks_split.upload(T.tasks_split);
    m_split = T.m_split;
    nsplit_lo = T.nsplit_lo;
    nsplit = (int)T.tasks_split.size();
}

LegArgs LegendreDev::args() const {
    LegArgs A;
    A.lmax = lmax;
    A.npair_pad = npair_pad;
    A.R = R;
    A.x = x.get();
    A.ls = ls.get();
    A.seedc = seedc.get();
    A.seedp = seedp.get();
    A.alpha = alpha.get();
    A.wg = 0;
    A.uni = uniform_start ? 1 : 0;
    return A;
}

LegArgs LegendreDev::args_synth() const {
    LegArgs A = args();
    A.R = Rs;
    A.wg = synth_wg ? 1 : 0;
    A.uni = uniform_start ? 1 : 0;
    return A;
}

void Legendre2Dev::upload(const Legendre2Tables& T) {
    lmax = T.lmax;
    npair_pad = T.npair_pad;
    R = T.R;
    nchunk = T.nchunk;
    ntasks = (int)T.tasks.size();
    seed.upload(T.seed);
    ls.upload(T.ls);
    alpha.upload(T.alpha);
    beta.upload(T.beta);
    cnorm.upload(T.cnorm);
    tasks.upload(T.tasks);
    // (alpha, beta)_l (-1)^(l - 1 - l0), interleaved (Leg2Args::abs_)
    std::vector<double> ab(2 * T.alpha.size(), 0.0);
    for (int m = 0; m <= T.lmax; ++m) {
        const int l0 = std::max(m, 2);
        const int64_t mo = moffp(T.lmax, m) - m;
        for (int l = l0 + 1; l <= T.lmax + 1; ++l) {
            const double sg = ((l - 1 - l0) & 1) ? -1.0 : 1.0;
            ab[2 * (mo + l)] = sg * T.alpha[mo + l];
            ab[2 * (mo + l) + 1] = sg * T.beta[mo + l];
        }
    }
    abs_.upload(ab);
    lw_chunk.upload(T.lw_chunk);
}

ShtPlan::ShtPlan(int nside, int lmax, const std::vector<int>& rings, const double* wring, int max_maps, bool pol)
    : max_maps_(max_maps), pol_(pol) {
    CMDR_REQUIRE(nside >= 1 && (nside & (nside - 1)) == 0, "nside must be a power of two");
    CMDR_REQUIRE(lmax >= 0, "lmax must be >= 0");
    CMDR_REQUIRE(max_maps >= 1, "max_maps must be >= 1");
    T_.build(nside, lmax, rings, wring, max_maps, pol);
    leg_.upload(T_.leg);
    std::vector<RingDev> rd(T_.ring.npair);
    for (int p = 0; p < T_.ring.npair; ++p) {
        const RingPairDesc& d = T_.ring.pairs[p];
        RingDev& r = rd[p];
        r.nphi = d.nphi;
        r.log2M = d.log2M;
        r.bluestein = d.bluestein;
        r.mmax_eff = d.mmax_eff;
        r.startN = d.startN;
        r.startS = d.startS;
        r.phi0 = d.phi0;
        r.wgt = d.wgt;
        r.chirp_off = d.chirp_off;
        r.ring = d.ring;
        r.split = d.split;
        r.log2T = d.log2T;
        r.that_off = d.that_off;
    }
    rings_.upload(rd);
    auto up = [](const std::vector<std::vector<int>>& src, std::vector<DevBuf<int>>& dst, std::vector<int>& n) {
        dst.resize(src.size());
        n.resize(src.size());
        for (size_t c = 0; c < src.size(); ++c) {
            n[c] = (int)src[c].size();
            if (n[c]) dst[c].upload(src[c]);
        }
    };
    up(T_.ring.classes, cls_, ncls_);
    if (T_.ring.that_elems) {
        up(T_.ring.classes_t, cls_t_, ncls_t_);
        up(T_.ring.classes_tb, cls_tb_, ncls_tb_);
        std::vector<std::vector<int>> ts(T_.ring.classes_t.size());      // the Toeplitz pairs by circulant class
        for (size_t c = 0; c < ts.size(); ++c)
            for (int p : T_.ring.classes_t[c]) if (T_.ring.pairs[p].log2T) ts[c].push_back(p);
        up(ts, cls_ts_, ncls_ts_);
        for (int p = 0; p < T_.ring.npair; ++p) if (rd[p].log2T) rd[p].mmax_eff *= 2;
        rings_t2_.upload(rd);
    }
    tw_.upload(T_.ring.twiddle);
    chirp_.upload(T_.ring.chirp);
    ast_.alloc((size_t)max_maps * leg_.tri_elems());
    ph_.alloc((size_t)max_maps * leg_.ph_elems());
    part_.alloc((size_t)max_maps * part_map_stride());
    if (T_.ring.nsplit) ring_scratch_.alloc((size_t)max_maps * T_.ring.nsplit * T_.ring.split_line * 2);
    if (pol) {
        CMDR_REQUIRE(max_maps >= 2, "a polarised plan needs max_maps >= 2 (Q and U phases)");
        leg2_.upload(T_.leg2);
        const int npol = max_maps / 2;
        st2_.alloc((size_t)npol * leg2_.tri4());
        part2_.alloc((size_t)npol * part2_pol_stride());
        st2_.zero();
        part2_.zero();
    }
    // never-written entries ((m, pair) beyond mlim, l below a task's start) must read as zero forever
    ast_.zero();
    ph_.zero();
    part_.zero();
    CMDR_HIP_CHECK(hipDeviceSynchronize());
}

void ShtPlan::synth_from_stream(int nmaps, hipStream_t s) {
    launch_leg_synth(leg_.args_synth(), leg_.tasks_s.get(), leg_.ntasks_s, ast_.get(), ph_.get(), leg_.ph_elems(), nmaps,
                     s);
}

bool ShtPlan::can_prep() const { return leg_synth_can_prep(leg_.args_synth()); }
void ShtPlan::synth_from_prep(const PrepDev& prep, int nmaps, hipStream_t s) {
    CMDR_REQUIRE(can_prep(), "this plan's synthesis reads its coefficients from the stream");
    launch_leg_synth(leg_.args_synth(), leg_.tasks_s.get(), leg_.ntasks_s, ast_.get(), ph_.get(), leg_.ph_elems(), nmaps,
                     s, -1, &prep);
}

void ShtPlan::rings(int mode, double* d_map, int64_t map_stride, const double* const* d_mul, bool weighted,
                    int nmaps, hipStream_t s, const cd* that) {
    const bool tz = mode == 2 && that && T_.ring.that_elems;      // classes follow the LDS image each pair then needs
    const std::vector<DevBuf<int>>& cls = tz ? cls_t_ : cls_;
    const std::vector<int>& ncls = tz ? ncls_t_ : ncls_;
    for (size_t c = 0; c < cls.size(); ++c) {
        if (!ncls[c]) continue;
        launch_ring(mode, rings_.get(), cls[c].get(), ncls[c], (int)c, ph_.get(), leg_.ph_elems(),
                    T_.lmax + 1, d_map, map_stride, d_mul, weighted ? 1 : 0,
                    reinterpret_cast<const cd*>(tw_.get()), T_.ring.log2Mmax,
                    reinterpret_cast<const cd*>(chirp_.get()), reinterpret_cast<cd*>(ring_scratch_.get()),
                    (int64_t)T_.ring.nsplit * T_.ring.split_line, T_.ring.split_line, nmaps, s, tz ? that : nullptr,
                    T_.ring.that_elems);
    }
}

void ShtPlan::toeplitz_build(const std::vector<const double*>& mul_host, DevBuf<cd>& out, hipStream_t s) {
    const int64_t ne = T_.ring.that_elems;
    if (!ne || mul_host.empty()) { out.release(); return; }
    out.alloc((size_t)ne * mul_host.size());
    // t_d = sum_k mul_k e^{-i d phi_k}, d <= 2 mmax: the analysis ring transform (mode 1) of the multiplier map itself,
    // stored like phases in a scratch array of 2 lmax + 1 rows
    int rows = 1;
    for (const RingPairDesc& d : T_.ring.pairs) if (d.log2T) rows = std::max(rows, 2 * d.mmax_eff + 1);
    DevBuf<double> td((size_t)rows * leg_.npair_pad * 4);
    const cd* tw = reinterpret_cast<const cd*>(tw_.get());
    for (size_t k = 0; k < mul_host.size(); ++k) {
        for (size_t c = 0; c < cls_tb_.size(); ++c)
            if (ncls_tb_[c])
                launch_ring(1, rings_t2_.get(), cls_tb_[c].get(), ncls_tb_[c], (int)c, td.get(), 0, rows,
                            const_cast<double*>(mul_host[k]), 0, nullptr, 0, tw, T_.ring.log2Mmax,
                            reinterpret_cast<const cd*>(chirp_.get()), nullptr, 0, 0, 1, s);
        for (size_t c = 0; c < cls_ts_.size(); ++c)
            if (ncls_ts_[c])
                launch_ring_toeplitz_spec(rings_.get(), cls_ts_[c].get(), ncls_ts_[c], (int)c, td.get(), rows,
                                          out.get() + (int64_t)k * ne, tw, T_.ring.log2Mmax, s);
    }
    CMDR_HIP_CHECK(hipStreamSynchronize(s));   // td is released on return
}

void ShtPlan::synth_range(int k0, int n, int nbs, hipStream_t s) {
    launch_leg_synth(leg_.args_synth(), leg_.tasks_s.get(), leg_.ntasks_s, ast_.get() + 2 * k0,
                     ph_.get() + (int64_t)k0 * leg_.ph_elems(), leg_.ph_elems(), n, s, nbs);
}

void ShtPlan::rings_fused_range(int k0, int n, const double* const* d_mul, hipStream_t s) {
    const int64_t sms = (int64_t)T_.ring.nsplit * T_.ring.split_line;
    for (size_t c = 0; c < cls_.size(); ++c) {
        if (!ncls_[c]) continue;
        launch_ring(2, rings_.get(), cls_[c].get(), ncls_[c], (int)c, ph_.get() + (int64_t)k0 * leg_.ph_elems(),
                    leg_.ph_elems(), T_.lmax + 1, nullptr, 0, d_mul + k0, 0,
                    reinterpret_cast<const cd*>(tw_.get()), T_.ring.log2Mmax,
                    reinterpret_cast<const cd*>(chirp_.get()),
                    reinterpret_cast<cd*>(ring_scratch_.get()) + (int64_t)k0 * sms, sms, T_.ring.split_line, n, s);
    }
}

void ShtPlan::adjoint_range(int k0, int n, hipStream_t s) {
    launch_leg_adj(leg_.args(), leg_.tasks.get(), leg_.ntasks, ph_.get() + (int64_t)k0 * leg_.ph_elems(), leg_.ph_elems(),
                   part_.get() + (int64_t)k0 * part_map_stride(), part_map_stride(), leg_.tri_elems(), n, false, s);
}

void ShtPlan::adjoint_to_partials(int nmaps, bool square, hipStream_t s, const std::function<void(int)>& between) {
    launch_leg_adj(leg_.args(), leg_.tasks.get(), leg_.ntasks, ph_.get(), leg_.ph_elems(), part_.get(),
                   part_map_stride(), leg_.tri_elems(), nmaps, square, s, between);
}

void ShtPlan::adjoint_half_to_partials(int nmaps, int half, hipStream_t s) {
    const WaveTask* t = leg_.tasks_split.get() + (half == 0 ? 0 : leg_.nsplit_lo);
    const int n = half == 0 ? leg_.nsplit_lo : leg_.nsplit - leg_.nsplit_lo;
    launch_leg_adj(leg_.args(), t, n, ph_.get(), leg_.ph_elems(), part_.get(), part_map_stride(), leg_.tri_elems(), nmaps,
                   false, s, nullptr);
}

void ShtPlan::alm2map(const double* d_alm, int64_t alm_stride, double* d_map, int64_t map_stride, int nmaps,
                      bool weighted, hipStream_t s) {
    for (int i0 = 0; i0 < nmaps; i0 += max_maps_) {
        const int nb = std::min(max_maps_, nmaps - i0);
        launch_alm_to_stream(d_alm + i0 * alm_stride, alm_stride, ast_.get(), leg_.cnorm.get(), T_.lmax, nb, s);
        synth_from_stream(nb, s);
        rings(0, d_map + i0 * map_stride, map_stride, nullptr, weighted, nb, s);
    }
}

void ShtPlan::map2alm(const double* d_map, int64_t map_stride, double* d_alm, int64_t alm_stride, int nmaps,
                      bool weighted, hipStream_t s) {
    for (int i0 = 0; i0 < nmaps; i0 += max_maps_) {
        const int nb = std::min(max_maps_, nmaps - i0);
        rings(1, const_cast<double*>(d_map) + i0 * map_stride, map_stride, nullptr, weighted, nb, s);
        adjoint_to_partials(nb, false, s);
        launch_part_to_alm(part_.get(), part_map_stride(), leg_.tri_elems(), leg_.nchunk, d_alm + i0 * alm_stride,
                           alm_stride, leg_.cnorm.get(), T_.lmax, nb, s, leg_.lw_chunk.get());
    }
}

std::vector<double> ShtPlan::pixel_weights() const {
    std::vector<double> w((size_t)T_.ring.npix_local, 0.0);
    for (const RingPairDesc& d : T_.ring.pairs) {
        for (int k = 0; k < d.nphi; ++k) {
            w[d.startN + k] = d.wgt;
            if (d.startS >= 0) w[d.startS + k] = d.wgt;
        }
    }
    return w;
}

Leg2Args ShtPlan_leg2_args(const Legendre2Dev& L, const double* x) {
    Leg2Args A;
    A.lmax = L.lmax;
    A.npair_pad = L.npair_pad;
    A.R = L.R;
    A.x = x;
    A.ls = L.ls.get();
    A.seed = L.seed.get();
    A.alpha = L.alpha.get();
    A.beta = L.beta.get();
    A.abs_ = L.abs_.get();
    return A;
}

void ShtPlan::synth2_from_stream(int npol, int kq0, hipStream_t s) {
    CMDR_REQUIRE(pol_, "plan was created without polarisation");
    launch_leg2_synth(ShtPlan_leg2_args(leg2_, leg_.x.get()), leg2_.tasks.get(), leg2_.ntasks, st2_.get(), npol, ph_.get(),
                      leg_.ph_elems(), kq0, s);
}

void ShtPlan::adjoint2_to_partials(int npol, int kq0, hipStream_t s) {
    CMDR_REQUIRE(pol_, "plan was created without polarisation");
    launch_leg2_adj(ShtPlan_leg2_args(leg2_, leg_.x.get()), leg2_.tasks.get(), leg2_.ntasks, ph_.get(), leg_.ph_elems(), kq0,
                    part2_.get(), part2_pol_stride(), leg2_.tri4(), npol, s);
}

void ShtPlan::alm2map_spin2(const double* d_E, const double* d_B, double* d_Q, double* d_U, bool weighted,
                            hipStream_t s) {
    CMDR_REQUIRE(pol_, "plan was created without polarisation");
    launch_alm2_to_stream(d_E, d_B, 0, st2_.get(), 1, leg2_.cnorm.get(), T_.lmax, s);
    synth2_from_stream(1, 0, s);
    // Q and U are two scalar-like maps for the ring stage (phase maps 0 and 1)
    rings(0, d_Q, d_U - d_Q, nullptr, weighted, 2, s);
}

void ShtPlan::map2alm_spin2(const double* d_Q, const double* d_U, double* d_E, double* d_B, bool weighted,
                            hipStream_t s) {
    CMDR_REQUIRE(pol_, "plan was created without polarisation");
    rings(1, const_cast<double*>(d_Q), d_U - d_Q, nullptr, weighted, 2, s);
    adjoint2_to_partials(1, 0, s);
    launch_part2_to_alm(part2_.get(), part2_pol_stride(), leg2_.tri4(), leg2_.nchunk, d_E, d_B, 0, leg2_.cnorm.get(),
                        T_.lmax, 1, s, leg2_.lw_chunk.get());
}

// share_in : every scalar column has the same input (d_in holds it once) and every (Q,U) pair the same (E,B) input
//            (once, after the scalar column): one synthesis instead of one per column.
// sum_out  : the scalar outputs are wanted summed (d_out holds one column), likewise the (E,B) outputs: one adjoint
//            (Yt is linear).
// Scalar columns share in pixel space -- the one synthesised map feeds nT analysis FFTs with their own multipliers, or
// the nT multiplied maps are summed before one analysis FFT -- so the ring stage does n + 1 transforms instead of 2 n.
// (Q,U) pairs share in phase space (slots copied before / summed after the fused ring stage).
void ShtPlan::sandwich(const double* d_in, double* d_out, const double* const* d_mul, int nT, int npol,
                       hipStream_t s, bool share_in, bool sum_out) {
    CMDR_REQUIRE(nT + 2 * npol <= max_maps_, "sandwich: more columns than the plan was sized for");
    CMDR_REQUIRE(npol == 0 || pol_, "plan was created without polarisation");
    CMDR_REQUIRE(!(share_in && sum_out), "sandwich: share_in and sum_out are exclusive");
    const int64_t na = nalm(), pe = leg_.ph_elems(), np = npix_local();
    const int nTin = share_in ? std::min(nT, 1) : nT, nPin = share_in ? std::min(npol, 1) : npol;
    const int nTout = sum_out ? std::min(nT, 1) : nT, nPout = sum_out ? std::min(npol, 1) : npol;
    const bool pixT = nT > 1 && (share_in || sum_out);   // scalar columns through pixel space
    if (nT) {
        launch_alm_to_stream(d_in, na, ast_.get(), leg_.cnorm.get(), T_.lmax, nTin, s);
        synth_from_stream(nTin, s);
    }
    if (npol) {
        launch_alm2_to_stream(d_in + nTin * na, d_in + (nTin + 1) * na, 2 * na, st2_.get(), nPin, leg2_.cnorm.get(), T_.lmax, s);
        synth2_from_stream(nPin, nT, s);
        if (nPin < npol) {
            launch_phase_share(ph_.get() + (int64_t)nT * pe, 2 * pe, npol, pe, false, s);         // Q slots
            launch_phase_share(ph_.get() + (int64_t)(nT + 1) * pe, 2 * pe, npol, pe, false, s);   // U slots
        }
    }
    if (!pixT) {
        rings(2, nullptr, 0, d_mul, false, nT + 2 * npol, s);
    } else {
        share_map_.ensure((size_t)np * (share_in ? 1 : nT));
        if (share_in) {
            rings(0, share_map_.get(), np, nullptr, false, 1, s);
            rings(1, share_map_.get(), 0, d_mul, false, nT, s);
        } else {
            rings(0, share_map_.get(), np, d_mul, false, nT, s);
            launch_phase_share(share_map_.get(), np, nT, np, true, s);
            rings(1, share_map_.get(), 0, nullptr, false, 1, s);
        }
        if (npol) rings_fused_range(nT, 2 * npol, d_mul, s);
    }
    if (nT) {
        adjoint_to_partials(nTout, false, s);
        launch_part_to_alm(part_.get(), part_map_stride(), leg_.tri_elems(), leg_.nchunk, d_out, na, leg_.cnorm.get(),
                           T_.lmax, nTout, s, leg_.lw_chunk.get());
    }
    if (npol) {
        if (nPout < npol) {
            launch_phase_share(ph_.get() + (int64_t)nT * pe, 2 * pe, npol, pe, true, s);
            launch_phase_share(ph_.get() + (int64_t)(nT + 1) * pe, 2 * pe, npol, pe, true, s);
        }
        adjoint2_to_partials(nPout, nT, s);
        launch_part2_to_alm(part2_.get(), part2_pol_stride(), leg2_.tri4(), leg2_.nchunk, d_out + nTout * na,
                            d_out + (nTout + 1) * na, 2 * na, leg2_.cnorm.get(), T_.lmax, nPout, s, leg2_.lw_chunk.get());
    }
}

}  // namespace cmdr
