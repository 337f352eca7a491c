// TEST INFRASTRUCTURE ONLY: single-thread host emulation of the libcmdr_hip kernel *bodies*
// (commander_amd/csrc/kernels_body.hpp) driven by the real plan tables (plan_tables.cpp).
// It lets the CPU-only test tier check index maps, seeds, FFT/Bluestein algebra and normalisations against the
// oracle on a box without a GPU.  It is never linked into libcmdr_hip.so and is not a fallback path.
#include <cstring>
#include <vector>

#include "kernels_body.hpp"
#include "plan_tables.hpp"

using namespace cmdr;

namespace {

LegArgs leg_args(const LegendreTables& T) {
    LegArgs A;
    A.lmax = T.lmax;
    A.npair_pad = T.npair_pad;
    A.R = T.R;
    A.x = T.x.data();
    A.ls = T.ls.data();
    A.seedc = T.seedc.data();
    A.seedp = T.seedp.data();
    A.alpha = T.alpha.data();
    return A;
}

template <int R>
void synth_all(const LegendreTables& T, const double* ast, double* ph) {
    LegArgs A = leg_args(T);
    for (const WaveTask& t : T.tasks)
        for (int lane = 0; lane < 64; ++lane) leg_synth_lane<R>(A, ast, ph, t.m, t.chunk, t.lw, t.lAend, lane);
}

template <int R, bool SQ>
void adj_all(const LegendreTables& T, const double* ph, double* part, int64_t pcs) {
    LegArgs A = leg_args(T);
    const int lmax = T.lmax;
    for (const WaveTask& t : T.tasks) {
        std::vector<AdjLane<R>> S(64);
        for (int lane = 0; lane < 64; ++lane) leg_adj_load<R, SQ>(A, ph, t.m, t.chunk, lane, S[lane]);
        const int64_t mo = d_moffp(lmax, t.m);
        const double* al = A.alpha + (mo - t.m);
        double* out = part + t.chunk * pcs + 2 * (mo - t.m);
        for (int l0 = t.lw; l0 <= lmax; l0 += kAdjL_) {
            double wl[16 * 65];
            for (int lane = 0; lane < 64; ++lane) {
                double v[16];
                if (l0 < t.lAend) leg_adj_group<R, SQ, true>(A, al, l0, S[lane], v);
                else leg_adj_group<R, SQ, false>(A, al, l0, S[lane], v);
                for (int j = 0; j < 16; ++j) wl[j * 65 + lane] = v[j];
            }
            // same summation order as the device: 4 quarter sums of 16, then (q0+q1)+(q2+q3) butterfly
            for (int col = 0; col < 16; ++col) {
                double q[4];
                for (int qt = 0; qt < 4; ++qt) {
                    double s = 0.0;
                    for (int i = 0; i < 16; ++i) s += wl[col * 65 + qt * 16 + i];
                    q[qt] = s;
                }
                const double s = (q[0] + q[1]) + (q[2] + q[3]);
                const int l = l0 + (col >> 1);
                if (l <= lmax) out[2 * l + (col & 1)] = s;
            }
        }
    }
}

RingDev to_dev(const RingPairDesc& d) {
    RingDev r;
    r.nphi = d.nphi; r.log2M = d.log2M; r.bluestein = d.bluestein; r.mmax_eff = d.mmax_eff;
    r.startN = d.startN; r.startS = d.startS; r.phi0 = d.phi0; r.wgt = d.wgt; r.chirp_off = d.chirp_off;
    r.ring = d.ring; r.pad = 0;
    return r;
}

}  // namespace

extern "C" {

// job: 0 YtW, 1 Y, 2 Yt, 3 WY ; rings NULL = all ; mul = optional per-pixel multiplier (local map layout)
int emul_sht(int job, int nside, int lmax, int nrings, const int* rings, const double* wring, double* alm,
             double* map, const double* mul, int R_override) {
    ShtTables T;
    std::vector<int> r;
    if (rings && nrings > 0) r.assign(rings, rings + nrings);
    T.build(nside, lmax, r, wring);
    if (R_override > 0) {
        std::vector<double> x(T.leg.x.begin(), T.leg.x.begin() + T.leg.npair);
        std::vector<double> sth(T.leg.sth.begin(), T.leg.sth.begin() + T.leg.npair);
        T.leg.build(lmax, x, sth, R_override);
    }
    const LegendreTables& L = T.leg;
    const bool synth = (job == 1 || job == 3), weighted = (job == 0 || job == 3);
    std::vector<double> ast(2 * ntrip(lmax), 0.0), ph((size_t)(lmax + 1) * L.npair_pad * 4, 0.0);
    std::vector<double> part((size_t)L.nchunk * 2 * ntrip(lmax), 0.0);
    const cd* tw = reinterpret_cast<const cd*>(T.ring.twiddle.data());
    const cd* chirp = reinterpret_cast<const cd*>(T.ring.chirp.data());
    std::vector<cd> buf((size_t)1 << T.ring.log2Mmax);
    const FftCtx c{0, 1};
    if (synth) {
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax + 1; ++l) alm_to_stream_elem(alm, ast.data(), L.cnorm.data(), lmax, m, l);
        if (L.R == 1) synth_all<1>(L, ast.data(), ph.data());
        else if (L.R == 2) synth_all<2>(L, ast.data(), ph.data());
        else synth_all<4>(L, ast.data(), ph.data());
        for (int p = 0; p < T.ring.npair; ++p) {
            const RingDev d = to_dev(T.ring.pairs[p]);
            ring_synth_lds(buf.data(), d, ph.data(), L.npair_pad, p, tw, T.ring.log2Mmax, chirp, c);
            const double wg = weighted ? d.wgt : 1.0;
            for (int k = 0; k < d.nphi; ++k) {
                map[d.startN + k] = buf[k].x * wg * (mul ? mul[d.startN + k] : 1.0);
                if (d.startS >= 0) map[d.startS + k] = buf[k].y * wg * (mul ? mul[d.startS + k] : 1.0);
            }
        }
    } else {
        for (int p = 0; p < T.ring.npair; ++p) {
            const RingDev d = to_dev(T.ring.pairs[p]);
            const double wg = weighted ? d.wgt : 1.0;
            for (int k = 0; k < d.nphi; ++k) {
                buf[k].x = map[d.startN + k] * wg * (mul ? mul[d.startN + k] : 1.0);
                buf[k].y = d.startS >= 0 ? map[d.startS + k] * wg * (mul ? mul[d.startS + k] : 1.0) : 0.0;
            }
            ring_anal_lds(buf.data(), d, tw, T.ring.log2Mmax, chirp, c);
            ring_store_phases(buf.data(), d, ph.data(), L.npair_pad, p, c);
        }
        const int64_t pcs = 2 * ntrip(lmax);
        if (L.R == 1) adj_all<1, false>(L, ph.data(), part.data(), pcs);
        else if (L.R == 2) adj_all<2, false>(L, ph.data(), part.data(), pcs);
        else adj_all<4, false>(L, ph.data(), part.data(), pcs);
        for (int m = 0; m <= lmax; ++m)
            for (int l = m; l <= lmax; ++l)
                part_to_alm_elem(part.data(), pcs, L.nchunk, alm, L.cnorm.data(), lmax, m, l);
    }
    return 0;
}

// Fused ring stage check: phases(alm) -> pixels*mul -> phases -> alm  ==  Yt diag(mul) Y alm
int emul_fused(int nside, int lmax, const double* alm_in, const double* mul, double* alm_out) {
    ShtTables T;
    T.build(nside, lmax, {}, nullptr);
    const LegendreTables& L = T.leg;
    std::vector<double> ast(2 * ntrip(lmax), 0.0), ph((size_t)(lmax + 1) * L.npair_pad * 4, 0.0);
    std::vector<double> part((size_t)L.nchunk * 2 * ntrip(lmax), 0.0);
    const cd* tw = reinterpret_cast<const cd*>(T.ring.twiddle.data());
    const cd* chirp = reinterpret_cast<const cd*>(T.ring.chirp.data());
    std::vector<cd> buf((size_t)1 << T.ring.log2Mmax);
    const FftCtx c{0, 1};
    for (int m = 0; m <= lmax; ++m)
        for (int l = m; l <= lmax + 1; ++l) alm_to_stream_elem(alm_in, ast.data(), L.cnorm.data(), lmax, m, l);
    if (L.R == 1) synth_all<1>(L, ast.data(), ph.data());
    else if (L.R == 2) synth_all<2>(L, ast.data(), ph.data());
    else synth_all<4>(L, ast.data(), ph.data());
    for (int p = 0; p < T.ring.npair; ++p) {
        const RingDev d = to_dev(T.ring.pairs[p]);
        ring_synth_lds(buf.data(), d, ph.data(), L.npair_pad, p, tw, T.ring.log2Mmax, chirp, c);
        for (int k = 0; k < d.nphi; ++k) {
            buf[k].x *= mul[d.startN + k];
            buf[k].y = d.startS >= 0 ? buf[k].y * mul[d.startS + k] : 0.0;
        }
        ring_anal_lds(buf.data(), d, tw, T.ring.log2Mmax, chirp, c);
        ring_store_phases(buf.data(), d, ph.data(), L.npair_pad, p, c);
    }
    const int64_t pcs = 2 * ntrip(lmax);
    if (L.R == 1) adj_all<1, false>(L, ph.data(), part.data(), pcs);
    else if (L.R == 2) adj_all<2, false>(L, ph.data(), part.data(), pcs);
    else adj_all<4, false>(L, ph.data(), part.data(), pcs);
    for (int m = 0; m <= lmax; ++m)
        for (int l = m; l <= lmax; ++l)
            part_to_alm_elem(part.data(), pcs, L.nchunk, alm_out, L.cnorm.data(), lmax, m, l);
    return 0;
}

int64_t emul_npix_local(int nside, int nrings, const int* rings) {
    RingTables R;
    std::vector<int> r(rings, rings + nrings), mlim(nrings, 0);
    R.build(nside, 0, r, nullptr, mlim);
    return R.npix_local;
}

}  // extern "C"
