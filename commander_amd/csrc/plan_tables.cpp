// Host-side SHT plan tables (no HIP calls here; also compiled into the host-emulation test library).
#include "plan_tables.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <map>

#include "common_host.hpp"

namespace cmdr {

static inline double eps_lm(int l, int m) {
    const double dl = (double)l, dm = (double)m;
    return std::sqrt((dl * dl - dm * dm) / (4.0 * dl * dl - 1.0));
}

// |mu| must reach 2^kStartExp before a (m, pair) column is switched on; everything below is dropped.  2^-100 = 8e-31 of
// the O(1) plateau the recursion grows into: fourteen orders of magnitude below the fp64 rounding of the sums these
// terms would enter (the first version waited only for representability, 2^-280, and ran 5 % more steps; every parity
// test holds at either value).  CMDR_START_EXP overrides it for experiments.
static const int kStartExp = [] { const char* e = std::getenv("CMDR_START_EXP"); return e ? std::atoi(e) : -100; }();

void LegendreTables::build(int lmax_, const std::vector<double>& x_, const std::vector<double>& sth_, int R_,
                           int Rs_, int nthreads, const std::vector<int>* mlim_in) {
    lmax = lmax_;
    npair = (int)x_.size();
    R = R_;
    Rs = Rs_;
    const int per = kWave * R;
    const int padto = kWave * std::max(R, Rs);
    npair_pad = (npair + padto - 1) / padto * padto;
    nchunk = npair_pad / per;
    nchunk_s = npair_pad / (kWave * Rs);
    x.assign(npair_pad, 0.0);
    sth.assign(npair_pad, 1.0);
    mlim.assign(npair_pad, -1);
    for (int p = 0; p < npair; ++p) {
        x[p] = x_[p];
        sth[p] = sth_[p];
        mlim[p] = mlim_in ? (*mlim_in)[p] : mlim_spin0(lmax, sth_[p]);
    }
    const int nm = lmax + 1;
    alpha.assign(ntrip(lmax), 0.0);
    cnorm.assign(ntrip(lmax), 0.0);
    ls.assign((size_t)nm * npair_pad, kLsNever);
    seedc.assign((size_t)nm * npair_pad, 0.0);
    seedp.assign((size_t)nm * npair_pad, 0.0);

    std::vector<double> logpref(nm);
    logpref[0] = -0.5 * std::log(4.0 * kPi);
    for (int m = 1; m <= lmax; ++m) logpref[m] = logpref[m - 1] + 0.5 * std::log((2.0 * m + 1.0) / (2.0 * m));

    const bool want_uniform = [] { const char* e = std::getenv("CMDR_UNIFORM_START"); return !e || std::atoi(e) != 0; }();
    std::atomic<bool> uniform_fail{false};
    host_parallel_for(nm, [&](int mi) {
        // interleave long and short columns for balance
        const int m = (mi & 1) ? lmax - mi / 2 : mi / 2;
        const int64_t mo = moffp(lmax, m);
        double* al = alpha.data() + mo - m;   // al[l], l = m..lmax+1
        double* cn = cnorm.data() + mo - m;
        // lambda_l = c_l mu_l ; mu_l = alpha_l x mu_{l-1} - mu_{l-2}
        cn[m] = 1.0;
        al[m] = 0.0;
        double c2 = 1.0, c1 = 1.0;  // c_{l-2}, c_{l-1}
        if (m + 1 <= lmax + 1) {
            al[m + 1] = 1.0 / eps_lm(m + 1, m);
            cn[m + 1] = 1.0;
        }
        for (int l = m + 2; l <= lmax + 1; ++l) {
            const double el = eps_lm(l, m), el1 = eps_lm(l - 1, m);
            const double c = c2 * el1 / el;
            al[l] = c1 / (el * c);
            cn[l] = c;
            c2 = c1;
            c1 = c;
        }
        cn[lmax + 1] = 0.0;  // pad entry carries no signal
        int* lsm = ls.data() + (size_t)m * npair_pad;
        double* scm = seedc.data() + (size_t)m * npair_pad;
        double* spm = seedp.data() + (size_t)m * npair_pad;
        for (int p = 0; p < npair; ++p) {
            if (m > mlim[p]) continue;
            const double xx = x[p];
            const double l2 = (logpref[m] + (m > 0 ? (double)m * std::log(sth[p]) : 0.0)) / M_LN2;
            const double fl = std::floor(l2);
            long e = (long)fl;
            double lc = std::exp2(l2 - fl), lp = 0.0;
            if (m & 1) lc = -lc;
            int l = m;
            for (;;) {
                int ex;
                (void)std::frexp(std::max(std::fabs(lc), std::fabs(lp)), &ex);
                if (e + ex >= kStartExp) {
                    lsm[p] = l;
                    scm[p] = std::ldexp(lc, (int)e);
                    spm[p] = std::ldexp(lp, (int)e);
                    break;
                }
                if (l == lmax) break;
                const double ln = al[l + 1] * xx * lc - lp;
                lp = lc;
                lc = ln;
                ++l;
                if (std::fabs(lc) > 0x1p+300) {
                    lc *= 0x1p-300;
                    lp *= 0x1p-300;
                    e += 300;
                }
            }
        }
        // Uniform starts: the 64 ring pairs of a lane block (one wavefront column of every kernel) all switch on at the
        // same l -- the first l == m (mod 32) at or below the block's earliest start -- with their true, still tiny
        // mu there as seeds (representable: a block spans a few hundred binary orders at most).  The kernels then
        // inject seeds once per block, at a group boundary, instead of testing every lane at every l while the lanes
        // of a wave trickle in.  A lane whose seed would underflow keeps its own start and clears `uniform_start`.
        if (want_uniform) {
            for (int b0 = 0; b0 < npair; b0 += kWave) {
                const int b1 = std::min(b0 + kWave, npair);
                int lo = kLsNever;
                for (int p = b0; p < b1; ++p) lo = std::min(lo, lsm[p]);
                if (lo == kLsNever) continue;
                const int s0 = m + ((lo - m) / 32) * 32;
                for (int p = b0; p < b1; ++p) {
                    if (lsm[p] == kLsNever || lsm[p] == s0) continue;
                    const double xx = x[p];
                    const double l2 = (logpref[m] + (m > 0 ? (double)m * std::log(sth[p]) : 0.0)) / M_LN2;
                    const double fl = std::floor(l2);
                    long e = (long)fl;
                    double lc = std::exp2(l2 - fl), lp = 0.0;
                    if (m & 1) lc = -lc;
                    for (int l = m; l < s0; ++l) {
                        const double ln = al[l + 1] * xx * lc - lp;
                        lp = lc;
                        lc = ln;
                        if (std::fabs(lc) > 0x1p+300) {
                            lc *= 0x1p-300;
                            lp *= 0x1p-300;
                            e += 300;
                        }
                    }
                    int exc, exp_;
                    (void)std::frexp(lc, &exc);
                    (void)std::frexp(lp, &exp_);
                    if (lc == 0.0 || e + exc < -960 || (lp != 0.0 && e + exp_ < -960)) {
                        uniform_fail.store(true, std::memory_order_relaxed);
                        continue;
                    }
                    lsm[p] = s0;
                    scm[p] = std::ldexp(lc, (int)e);
                    spm[p] = std::ldexp(lp, (int)e);
                }
            }
        }
    }, nthreads);
    uniform_start = want_uniform && !uniform_fail.load();
    // ---- task lists
    auto make_tasks = [&](int Rt, std::vector<std::vector<WaveTask>>& out) {
        const int pr = kWave * Rt, nch = npair_pad / pr;
        out.assign(nm, {});
        for (int m = 0; m < nm; ++m) {
            const int* lsm = ls.data() + (size_t)m * npair_pad;
            for (int ch = 0; ch < nch; ++ch) {
                int lo = kLsNever, hi = -1;
                for (int p = ch * pr; p < (ch + 1) * pr; ++p) {
                    const int v = lsm[p];
                    if (v == kLsNever) continue;
                    lo = std::min(lo, v);
                    hi = std::max(hi, v);
                }
                WaveTask t;
                t.m = m;
                t.chunk = ch;
                if (hi < 0) {
                    // no pair of the chunk reaches the start threshold by lmax.  Inside the (m, ring) cut the task is
                    // kept with an empty l range: the ring stage reads every entry with m <= mlim, so the synthesis
                    // must write its zeros (the adjoint of such a task writes nothing; its columns are skipped)
                    bool inside = false;
                    for (int p = ch * pr; p < (ch + 1) * pr; ++p) inside = inside || m <= mlim[p];
                    if (!inside) continue;
                    t.lw = t.lAend = lmax + 1 + ((lmax + 1 - m) & 1);
                    out[m].push_back(t);
                    continue;
                }
                t.lw = lo - ((lo - m) & 1);
                int a = hi + 1;
                a += (a - m) & 1;
                t.lAend = a;
                out[m].push_back(t);
            }
        }
    };
    std::vector<std::vector<WaveTask>> tm;
    make_tasks(R, tm);
    tasks.clear();
    for (int m = 0; m < nm; ++m) tasks.insert(tasks.end(), tm[m].begin(), tm[m].end());
    lw_chunk.assign((size_t)nm * nchunk, lmax + 2);     // partial-column entries below it are never written (stay zero)
    for (const WaveTask& t : tasks) lw_chunk[(size_t)t.m * nchunk + t.chunk] = t.lw;
    // longest first == smallest lw first (all columns end at lmax)
    std::stable_sort(tasks.begin(), tasks.end(), [](const WaveTask& a, const WaveTask& b) { return a.lw < b.lw; });
    while (tasks.size() % 4) { WaveTask t; t.m = 0; t.chunk = -1; t.lw = lmax + 1; t.lAend = lmax + 1; tasks.push_back(t); }
    group = 4;
    {   // two halves of equal work by m (work of a task ~ its l range)
        double tot = 0.0, acc = 0.0;
        std::vector<double> wm(nm, 0.0);
        for (const WaveTask& t : tasks) if (t.chunk >= 0) { wm[t.m] += (double)std::max(0, lmax + 1 - t.lw); tot += (double)std::max(0, lmax + 1 - t.lw); }
        m_split = nm;
        for (int m = 0; m < nm; ++m) { acc += wm[m]; if (acc >= 0.5 * tot) { m_split = m + 1; break; } }
        tasks_split.clear();
        for (int half = 0; half < 2; ++half) {
            for (const WaveTask& t : tasks)          // `tasks` is already sorted longest first
                if (t.chunk >= 0 && (half == 0 ? t.m < m_split : t.m >= m_split)) tasks_split.push_back(t);
            while (tasks_split.size() % 4) { WaveTask t; t.m = 0; t.chunk = -1; t.lw = lmax + 1; t.lAend = lmax + 1; tasks_split.push_back(t); }
            if (half == 0) nsplit_lo = (int)tasks_split.size();
        }
    }
    if (std::getenv("CMDR_DEBUG_PLAN")) {
        double slots = 0, slotsA = 0, steps = 0;
        for (const WaveTask& t : tasks) if (t.chunk >= 0) { slots += (double)(lmax - t.lw + 1) * kWave * R; slotsA += (double)(std::min(t.lAend, lmax + 1) - t.lw) * kWave * R; }
        for (int m = 0; m < nm; ++m) for (int p = 0; p < npair; ++p) { const int v = ls[(size_t)m * npair_pad + p]; if (v != kLsNever) steps += lmax - v + 1; }
        std::fprintf(stderr, "[cmdr] legendre plan lmax=%d npair=%d R=%d: %zu tasks, lane-slot steps %.4g (phase A %.4g), pruned steps %.4g (%.1f%%), uniform block starts: %s\n",
                     lmax, npair, R, tasks.size(), slots, slotsA, steps, 100.0 * steps / slots, uniform_start ? "yes" : "no");
    }
    make_tasks(Rs, tm);
    tasks_s.clear();
    synth_wg = false;
    bool want_wg = nchunk_s % 4 == 0;
    if (const char* e = std::getenv("CMDR_SYNTH_WG")) want_wg = want_wg && std::atoi(e) != 0;
    if (want_wg) {
        // workgroup = 4 adjacent chunks of one m: the coefficient stream of the column is staged through LDS once
        // per workgroup instead of being streamed through the scalar cache by every wave (k_leg_synth_wg)
        struct Grp { int lw; WaveTask t[4]; };
        std::vector<Grp> groups;
        for (int m = 0; m < nm; ++m)
            for (int g = 0; g < nchunk_s / 4; ++g) {
                Grp G;
                G.lw = lmax + 2;
                for (int k = 0; k < 4; ++k) { G.t[k].m = m; G.t[k].chunk = -1; G.t[k].lw = lmax + 2; G.t[k].lAend = lmax + 2; }
                bool any = false;
                for (const WaveTask& t : tm[m])
                    if (t.chunk / 4 == g) { G.t[t.chunk % 4] = t; G.lw = std::min(G.lw, t.lw); any = true; }
                if (any) groups.push_back(G);   // also groups of empty tasks: they write the zeros the ring stage reads
            }
        std::stable_sort(groups.begin(), groups.end(), [](const Grp& a, const Grp& b) { return a.lw < b.lw; });
        for (const Grp& G : groups) tasks_s.insert(tasks_s.end(), G.t, G.t + 4);
        synth_wg = true;
    } else {
        for (int m = 0; m < nm; ++m) tasks_s.insert(tasks_s.end(), tm[m].begin(), tm[m].end());
        std::stable_sort(tasks_s.begin(), tasks_s.end(), [](const WaveTask& a, const WaveTask& b) { return a.lw < b.lw; });
        while (tasks_s.size() % 4) { WaveTask t; t.m = 0; t.chunk = -1; t.lw = lmax + 1; t.lAend = lmax + 1; tasks_s.push_back(t); }
    }
    if (std::getenv("CMDR_DEBUG_PLAN"))
        std::fprintf(stderr, "[cmdr] synthesis plan: Rs=%d, %zu tasks, %s form\n", Rs, tasks_s.size(),
                     synth_wg ? "workgroup (LDS tiles; coefficients can be formed in the staging)" : "wave");
}

// --------------------------------------------------------------------------------------------- spin-2 tables
void Legendre2Tables::build(int lmax_, const std::vector<double>& x, const std::vector<double>& sth, int npair_pad_,
                            int R_, int nthreads, const std::vector<int>* mlim_in) {
    lmax = lmax_;
    npair_pad = npair_pad_;
    R = R_;
    const int npair = (int)x.size();
    const int per = kWave * R;
    nchunk = npair_pad / per;
    const int nm = lmax + 1;
    mlim.assign(npair_pad, -1);
    for (int p = 0; p < npair; ++p) mlim[p] = mlim_in ? (*mlim_in)[p] : mlim_spin2(lmax, sth[p], x[p]);
    alpha.assign(ntrip(lmax), 0.0);
    beta.assign(ntrip(lmax), 0.0);
    cnorm.assign(ntrip(lmax), 0.0);
    ls.assign((size_t)nm * npair_pad, kLsNever);
    seed.assign((size_t)nm * npair_pad * 4, 0.0);
    // CMDR_START_EXP2 overrides the spin-2 start threshold alone (tests: -280 = the round-2 tables)
    const int start_exp = [] { const char* e = std::getenv("CMDR_START_EXP2"); return e ? std::atoi(e) : kStartExp; }();
    const bool want_uniform = [] { const char* e = std::getenv("CMDR_UNIFORM_START"); return !e || std::atoi(e) != 0; }();
    std::atomic<bool> uniform_fail{false};
    host_parallel_for(nm, [&](int mi) {
        const int m = (mi & 1) ? lmax - mi / 2 : mi / 2;
        const int l0 = std::max(m, 2);
        if (l0 > lmax) return;
        const int64_t mo = moffp(lmax, m);
        double* al = alpha.data() + mo - m;   // al[l], l = m..lmax+1
        double* be = beta.data() + mo - m;
        double* cn = cnorm.data() + mo - m;
        auto C = [&](int l) {
            const double dl = l, dm = m;
            return std::sqrt((dl * dl - dm * dm) * (dl * dl - 4.0) / (dl * dl * (4.0 * dl * dl - 1.0)));
        };
        // c_{l0} = c_{l0+1} = 1 ; c_{l+1} = c_{l-1} C_l / C_{l+1} ; alpha_{l+1} = c_l / (C_{l+1} c_{l+1}) ;
        // beta_{l+1} = 2 alpha_{l+1} m / (l (l+1))
        std::vector<double> c(lmax + 3, 0.0);
        c[l0] = 1.0;
        if (l0 + 1 <= lmax + 1) c[l0 + 1] = 1.0;
        for (int l = l0 + 1; l <= lmax; ++l) c[l + 1] = c[l - 1] * C(l) / C(l + 1);
        for (int l = l0; l <= lmax; ++l) {
            const double a = c[l] / (C(l + 1) * c[l + 1]);
            al[l + 1] = a;
            be[l + 1] = 2.0 * a * (double)m / ((double)l * (l + 1.0));
        }
        for (int l = l0; l <= lmax; ++l) cn[l] = c[l];
        // seeds: both chains from their closed-form start, scaled recursion until either exceeds the start threshold
        // (2^kStartExp, as for the scalar tables: round 2 waited only for representability, 2^-280, and ran ~5 % more steps)
        int* lsm = ls.data() + (size_t)m * npair_pad;
        double* sd = seed.data() + (size_t)m * npair_pad * 4;
        struct Chains { double lc[2], lp[2]; long e; };
        auto init = [&](int p, Chains& S) {            // closed-form (+-2)lambda at l0, both chains on one exponent
            const double th = std::atan2(sth[p], x[p]), c2 = std::cos(0.5 * th), s2 = std::sin(0.5 * th);
            long e[2];
            S.lp[0] = S.lp[1] = 0.0;
            for (int ch = 0; ch < 2; ++ch) {
                const int s = ch == 0 ? 2 : -2;
                double l2, sign;
                if (m >= 2) {
                    double lg = 0.5 * (std::log(2.0 * m + 1.0) - std::log(4.0 * kPi) + std::lgamma(2.0 * m + 1.0) -
                                       std::lgamma(m + s + 1.0) - std::lgamma(m - s + 1.0));
                    lg += (m - s) * std::log(c2) + (m + s) * std::log(s2);
                    l2 = lg / M_LN2;
                    sign = (m & 1) ? -1.0 : 1.0;
                } else {   // m = 0, 1: Goldberg sum at l = 2
                    static const double fact[8] = {1, 1, 2, 6, 24, 120, 720, 5040};
                    const int l = 2;
                    const double pref = ((m & 1) ? -1.0 : 1.0) *
                                        std::sqrt(5.0 / (4 * kPi) * fact[l + m] * fact[l - m] / (fact[l + s] * fact[l - s]));
                    double acc = 0;
                    for (int r = 0; r <= l - s; ++r) {
                        const int k = r + s - m;
                        if (k < 0 || k > l + s) continue;
                        const double c1 = fact[l - s] / (fact[r] * fact[l - s - r]), cc2 = fact[l + s] / (fact[k] * fact[l + s - k]);
                        acc += c1 * cc2 * (((l - r - s) & 1) ? -1.0 : 1.0) * std::pow(s2, 2 * l - 2 * r - s + m) *
                               std::pow(c2, 2 * r + s - m);
                    }
                    const double v = pref * acc;
                    l2 = v == 0.0 ? -1e30 : std::log2(std::fabs(v));
                    sign = v < 0 ? -1.0 : 1.0;
                }
                if (l2 < -1e20) { S.lc[ch] = 0.0; e[ch] = 0; }
                else {
                    const double fl = std::floor(l2);
                    e[ch] = (long)fl;
                    S.lc[ch] = sign * std::exp2(l2 - fl);
                }
            }
            // bring both chains to the common (larger) exponent
            const long E = std::max(e[0], e[1]);
            for (int ch = 0; ch < 2; ++ch) {
                const long d = E - e[ch];
                if (d > 0) S.lc[ch] *= d > 2000 ? 0.0 : std::ldexp(1.0, (int)-d);
            }
            S.e = E;
        };
        auto step = [&](int p, int l, Chains& S) {     // l -> l + 1
            for (int ch = 0; ch < 2; ++ch) {
                const double t = al[l + 1] * x[p] + (ch == 0 ? be[l + 1] : -be[l + 1]);
                const double ln = t * S.lc[ch] - S.lp[ch];
                S.lp[ch] = S.lc[ch];
                S.lc[ch] = ln;
            }
            if (std::max(std::fabs(S.lc[0]), std::fabs(S.lc[1])) > 0x1p+300) {
                for (int ch = 0; ch < 2; ++ch) { S.lc[ch] *= 0x1p-300; S.lp[ch] *= 0x1p-300; }
                S.e += 300;
            }
        };
        auto store = [&](int p, int l, const Chains& S) {
            lsm[p] = l;
            sd[4 * p + 0] = std::ldexp(S.lc[0], (int)S.e);
            sd[4 * p + 1] = std::ldexp(S.lp[0], (int)S.e);
            sd[4 * p + 2] = std::ldexp(S.lc[1], (int)S.e);
            sd[4 * p + 3] = std::ldexp(S.lp[1], (int)S.e);
        };
        for (int p = 0; p < npair; ++p) {
            if (m > mlim[p]) continue;
            Chains S;
            init(p, S);
            int l = l0;
            for (;;) {
                int ex;
                (void)std::frexp(std::max(std::max(std::fabs(S.lc[0]), std::fabs(S.lp[0])),
                                          std::max(std::fabs(S.lc[1]), std::fabs(S.lp[1]))), &ex);
                if (S.e + ex >= start_exp) { store(p, l, S); break; }
                if (l == lmax) break;
                step(p, l, S);
                ++l;
            }
        }
        // Uniform starts (as LegendreTables::build): the 64 ring pairs of a lane block switch on at one l == l0 (mod 32),
        // the last such l at or below the block's earliest start, with their true, still tiny mu+- there as seeds; the
        // matrix-unit adjoint then injects seeds in one 32-l group per block.  A lane whose seeds would underflow keeps
        // its own start (the kernels handle per-lane starts in every group up to the block's last one).
        if (want_uniform) {
            for (int b0 = 0; b0 < npair; b0 += kWave) {
                const int b1 = std::min(b0 + kWave, npair);
                int lo = kLsNever;
                for (int p = b0; p < b1; ++p) lo = std::min(lo, lsm[p]);
                if (lo == kLsNever) continue;
                const int s0 = l0 + ((lo - l0) / 32) * 32;
                for (int p = b0; p < b1; ++p) {
                    if (lsm[p] == kLsNever || lsm[p] == s0) continue;
                    Chains S;
                    init(p, S);
                    for (int l = l0; l < s0; ++l) step(p, l, S);
                    bool ok = true;
                    for (int ch = 0; ch < 2; ++ch) {
                        int exc, exp_;
                        (void)std::frexp(S.lc[ch], &exc);
                        (void)std::frexp(S.lp[ch], &exp_);
                        if (S.lc[ch] == 0.0 || S.e + exc < -960 || (S.lp[ch] != 0.0 && S.e + exp_ < -960)) ok = false;
                    }
                    if (!ok) { uniform_fail.store(true, std::memory_order_relaxed); continue; }
                    store(p, s0, S);
                }
            }
        }
    }, nthreads);
    uniform_start = want_uniform && !uniform_fail.load();
    tasks.clear();
    for (int m = 0; m < nm; ++m) {
        const int l0 = std::max(m, 2);
        if (l0 > lmax) continue;
        const int* lsm = ls.data() + (size_t)m * npair_pad;
        for (int ch = 0; ch < nchunk; ++ch) {
            int lo = kLsNever, hi = -1;
            for (int p = ch * per; p < (ch + 1) * per; ++p) {
                const int v = lsm[p];
                if (v == kLsNever) continue;
                lo = std::min(lo, v);
                hi = std::max(hi, v);
            }
            WaveTask t;
            t.m = m;
            t.chunk = ch;
            if (hi < 0) {
                // no pair of the chunk reaches the start threshold by lmax.  Inside the (m, ring) cut the task is kept with
                // an empty l range: the ring stage reads every entry with m <= mlim, so the synthesis must write its zeros
                bool inside = false;
                for (int p = ch * per; p < (ch + 1) * per; ++p) inside = inside || m <= mlim[p];
                if (!inside) continue;
                t.lw = t.lAend = lmax + 1 + ((lmax + 1 - l0) & 1);
                tasks.push_back(t);
                continue;
            }
            t.lw = lo - ((lo - l0) & 1);           // pairs (l, l+1) start at l0-parity
            int a = hi + 1;
            a += (a - l0) & 1;
            t.lAend = a;
            tasks.push_back(t);
        }
    }
    lw_chunk.assign((size_t)nm * nchunk, lmax + 2);     // partial-column entries below it are never written (stay zero)
    for (const WaveTask& t : tasks) lw_chunk[(size_t)t.m * nchunk + t.chunk] = t.lw;
    std::stable_sort(tasks.begin(), tasks.end(), [](const WaveTask& a, const WaveTask& b) { return a.lw < b.lw; });
    while (tasks.size() % 4) { WaveTask t; t.m = 0; t.chunk = -1; t.lw = lmax + 1; t.lAend = lmax + 1; tasks.push_back(t); }
}

void ShtTables::build_spin2(int nthreads) {
    if (leg2.lmax == lmax) return;
    std::vector<double> x(leg.x.begin(), leg.x.begin() + leg.npair), sth(leg.sth.begin(), leg.sth.begin() + leg.npair);
    int R2 = 2;
    if (const char* e = std::getenv("CMDR_LEG2_R")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) R2 = v; }   // experiment
    while (R2 > 1 && leg.npair_pad % (kWave * R2) != 0) R2 >>= 1;
    // leg.mlim is already the merged cut on polarised plans (ShtTables::build)
    std::vector<int> ml(leg.mlim.begin(), leg.mlim.begin() + leg.npair);
    leg2.build(lmax, x, sth, leg.npair_pad, R2, nthreads, &ml);
}

static int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

static void host_fft_pow2(std::vector<std::complex<double>>& a, int sign) {
    const int n = (int)a.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1;
        for (int k = 0; k < half; ++k) {
            const double ang = sign * 2.0 * kPi * (double)k / (double)len;
            const std::complex<double> w(std::cos(ang), std::sin(ang));
            for (int i = k; i < n; i += len) {
                const auto u = a[i], v = a[i + half] * w;
                a[i] = u + v;
                a[i + half] = u - v;
            }
        }
    }
}

void RingTables::build(int nside_, int lmax_, const std::vector<int>& rings, const double* wring,
                       const std::vector<int>& mlim) {
    nside = nside_;
    lmax = lmax_;
    npair = (int)rings.size();
    pairs.assign(npair, RingPairDesc{});
    const double npix_full = 12.0 * nside * (double)nside;
    // local map = owned rings in ascending ring order: northern members ascending, then southern ascending
    int64_t off = 0;
    for (int p = 0; p < npair; ++p) {
        const RingInfo r = healpix_ring(nside, rings[p]);
        pairs[p].startN = off;
        off += r.nphi;
    }
    for (int p = npair - 1; p >= 0; --p) {
        if (rings[p] == 2 * nside) { pairs[p].startS = -1; continue; }
        const RingInfo r = healpix_ring(nside, rings[p]);
        pairs[p].startS = off;
        off += r.nphi;
    }
    npix_local = off;
    log2Mmax = 1;
    that_elems = 0;
    nsplit = 0;
    split_line = 0;
    constexpr int kMaxLog2M = 13;   // 8192 complex + padding = 147 KB of the 160 KB LDS
    int split_min_n = 1 << 30;      // test hook: split every ring at least this long
    if (const char* e = std::getenv("CMDR_RING_SPLIT_MIN_N")) { const int v = std::atoi(e); if (v >= 4) split_min_n = v; }
    std::map<int, int64_t> chirp_of;  // nphi -> offset (in complex units)
    chirp.clear();
    for (int p = 0; p < npair; ++p) {
        const RingInfo r = healpix_ring(nside, rings[p]);
        RingPairDesc& d = pairs[p];
        d.ring = rings[p];
        d.nphi = r.nphi;
        d.phi0 = r.phi0;
        d.mmax_eff = std::min(mlim[p], lmax);
        d.wgt = (wring ? wring[rings[p] - 1] : 1.0) * 4.0 * kPi / npix_full;
        const int n = r.nphi;
        // transform length: the whole ring, or (rings whose LDS image would exceed 2^kMaxLog2M complex, i.e. the
        // long cap rings of Nside 2048) two half-length pieces
        auto size_class = [](int nt, bool& p2) {
            p2 = (nt & (nt - 1)) == 0;
            int lg = 0;
            while ((1 << lg) < (p2 ? nt : 2 * nt - 1)) ++lg;
            return lg;
        };
        bool pow2;
        int nt = n, lg = size_class(n, pow2);
        d.split = 0;
        if (lg > kMaxLog2M || n >= split_min_n) {
            CMDR_REQUIRE(n % 2 == 0, "cannot split an odd-length ring");
            nt = n / 2;
            lg = size_class(nt, pow2);
            CMDR_REQUIRE(lg <= kMaxLog2M, "ring FFT too long (nside > 2048 is not supported)");
            d.split = ++nsplit;
            split_line = std::max(split_line, nt);
        }
        d.bluestein = pow2 ? 0 : 1;
        d.log2M = lg;
        auto it = chirp_of.find(n);
        if (it == chirp_of.end()) {
            // per-length table: rot_j = e^{i pi j/n} (j<n); Bluestein lengths add the chirp w_j = e^{i pi j^2/nt}
            // (j<nt) and the FFT_M^- of the conjugate chirp in bit-reversed order (M entries)
            const int64_t o = (int64_t)chirp.size() / 2;
            chirp_of[n] = o;
            for (int j = 0; j < n; ++j) {
                const double ang = kPi * (double)j / (double)n;
                chirp.push_back(std::cos(ang));
                chirp.push_back(std::sin(ang));
            }
            if (!pow2) {
                const int M = 1 << lg;
                std::vector<std::complex<double>> w(nt), c(M, 0.0);
                for (int j = 0; j < nt; ++j) {
                    const int64_t q = ((int64_t)j * j) % (2 * (int64_t)nt);
                    const double ang = kPi * (double)q / (double)nt;
                    w[j] = {std::cos(ang), std::sin(ang)};
                }
                c[0] = std::conj(w[0]);
                for (int j = 1; j < nt; ++j) { c[j] = std::conj(w[j]); c[M - j] = std::conj(w[j]); }
                host_fft_pow2(c, -1);
                for (int j = 0; j < nt; ++j) { chirp.push_back(w[j].real()); chirp.push_back(w[j].imag()); }
                for (int q = 0; q < M; ++q) {
                    const auto v = c[bitrev(q, lg)];
                    chirp.push_back(v.real());
                    chirp.push_back(v.imag());
                }
            }
            d.chirp_off = o;
        } else {
            d.chirp_off = it->second;
        }
        log2Mmax = std::max(log2Mmax, d.log2M);
        // Toeplitz form of the fused operator: worth it when its power-of-two circulant is no longer than the
        // Bluestein image (2 radix-2 FFTs instead of 4); split rings keep the pixel form
        d.log2T = 0;
        d.that_off = 0;
        if (d.bluestein && !d.split && d.mmax_eff >= 0) {
            int lt = 1;
            while ((1 << lt) < 4 * d.mmax_eff + 1) ++lt;
            if (lt <= d.log2M) {
                d.log2T = lt;
                d.that_off = that_elems;
                that_elems += (int64_t)1 << lt;
            }
        }
    }
    if (const char* e = std::getenv("CMDR_RING_TOEPLITZ")) if (std::atoi(e) == 0) { for (auto& d : pairs) d.log2T = 0; that_elems = 0; }
    if (std::getenv("CMDR_DEBUG_PLAN")) {
        int nt = 0;
        for (const RingPairDesc& d : pairs) nt += d.log2T != 0;
        std::fprintf(stderr, "[cmdr] ring plan nside=%d: %d pairs, %d split (line %d), log2Mmax %d, %d pairs in Toeplitz form (%lld spectrum entries per map)\n",
                     nside, npair, nsplit, split_line, log2Mmax, nt, (long long)that_elems);
    }
    classes.assign(log2Mmax + 1, {});
    // launch classes = LDS image size; all rings up to 4096 points share one launch (the few hundred workgroups per
    // size of the short rings fill the gaps of the long ones: 1.21 -> 1.18 ms per fused nine-map pass; CMDR_RING_MINCLASS)
    const int kMinClass = std::min(log2Mmax, [] { const char* e = std::getenv("CMDR_RING_MINCLASS"); return e ? std::atoi(e) : 12; }());
    classes_t.assign(log2Mmax + 1, {});
    classes_tb.assign(log2Mmax + 1, {});
    for (int p = 0; p < npair; ++p) {
        classes[std::max(pairs[p].log2M, kMinClass)].push_back(p);
        classes_t[std::max(pairs[p].log2T ? pairs[p].log2T : pairs[p].log2M, kMinClass)].push_back(p);
        if (pairs[p].log2T) classes_tb[std::max(pairs[p].log2M, kMinClass)].push_back(p);
    }
    const int Mmax = 1 << log2Mmax;
    twiddle.resize(Mmax);  // Mmax/2 complex
    for (int k = 0; k < Mmax / 2; ++k) {
        const double ang = 2.0 * kPi * (double)k / (double)Mmax;
        twiddle[2 * k] = std::cos(ang);
        twiddle[2 * k + 1] = std::sin(ang);
    }
}

void ShtTables::build(int nside_, int lmax_, const std::vector<int>& rings_in, const double* wring, int max_maps,
                      bool pol, int nthreads) {
    nside = nside_;
    lmax = lmax_;
    std::vector<int> rings = rings_in;
    if (rings.empty()) for (int i = 1; i <= 2 * nside; ++i) rings.push_back(i);
    std::sort(rings.begin(), rings.end());
    std::vector<double> x(rings.size()), sth(rings.size());
    for (size_t p = 0; p < rings.size(); ++p) {
        const RingInfo r = healpix_ring(nside, rings[p]);
        x[p] = r.z;
        sth[p] = r.sth;
    }
    const int np = (int)rings.size();
    // Adjoint: 4 ring pairs per lane (tasks of 256 pairs: the matrix-unit kernel keeps their phases in registers, the
    // VALU kernel amortises its cross-lane reduction) -- also on ring-sharded ranks with 256 or 512 pairs: 128-pair
    // tasks (R = 2, twice as many and half as long) measured 10 % slower there (cr_time_rank.py).  Synthesis: 2
    // pairs per lane (higher occupancy; its coefficients come through LDS in the workgroup form, which needs the
    // chunks of an m in groups of 4 -- 1 pair per lane where 2 would leave fewer than 4 chunks, e.g. 256-pair shards).
    // Measured with tools/cr_time.py and tools/cr_time_rank.py; tunable through CMDR_LEG_R / CMDR_LEG_RS.
    int R = 4, Rs = 2;
    while (R > 1 && np < 64 * R) R >>= 1;
    while (Rs > 1 && (np < 64 * Rs || ((np + 64 * R - 1) / (64 * R) * R / Rs) % 4 != 0)) Rs >>= 1;
    if (const char* e = std::getenv("CMDR_LEG_R")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) R = v; }
    if (const char* e = std::getenv("CMDR_LEG_RS")) { const int v = std::atoi(e); if (v == 1 || v == 2 || v == 4) Rs = v; }
    (void)max_maps;
    // One (m, ring pair) cut per plan.  The ring stage reads and rewrites every phase entry with m <= its cut for
    // every map slot, so each Legendre synthesis must write exactly that set: on polarised plans the T (spin-0)
    // tasks are built from the merged cut max(mlim_spin0, mlim_spin2) as well, otherwise the T slots would keep
    // analysis output of an earlier call in the entries mlim0 < m <= mlim2 (22 per pair at Nside 1024 / lmax 2000).
    std::vector<int> ml(np);
    for (int p = 0; p < np; ++p) {
        ml[p] = mlim_spin0(lmax, sth[p]);
        if (pol) ml[p] = std::max(ml[p], mlim_spin2(lmax, sth[p], x[p]));
    }
    leg.build(lmax, x, sth, R, Rs, nthreads, &ml);
    if (pol) build_spin2(nthreads);
    ring.build(nside, lmax, rings, wring, leg.mlim);
}

void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(kPi * (i + 0.75) / (n + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            const double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-16) break;
        }
        x[i] = z;
        x[n - 1 - i] = -z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

}  // namespace cmdr
