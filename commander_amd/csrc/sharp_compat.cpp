// The literal libsharp2 entry points of include/cmdr_sharp.h, implemented on top of the SHT-level C ABI.
#include "../../include/cmdr_sharp.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../include/cmdr_hip.h"

struct sharp_alm_info {
    int lmax;
    std::vector<int> ms;             // the m this rank owns, in the caller's order (comm_map_mod.f90:228-261)
    ptrdiff_t count;
    bool all_m;                      // ms = 0..lmax in order: the local layout is the full one
};

struct sharp_geom_info {
    int nside;
    std::vector<int> north;          // northern ring numbers owned (mirror implied)
    std::vector<double> weight;      // [2*nside] or empty
    ptrdiff_t npix;
    mutable std::map<int, cmdr_sht_plan*> plans;  // by 2 * lmax + pol
};

namespace {
[[noreturn]] void die(const char* what) {
    std::fprintf(stderr, "libcmdr_hip (sharp compat): %s\n", what);
    std::abort();
}

struct CommEntry { cmdr_sharp_allreduce_fn fn; void* user; };
std::map<int, CommEntry>& comms() { static std::map<int, CommEntry> c; return c; }

inline ptrdiff_t full_count(int lmax) { return (ptrdiff_t)(lmax + 1) * (lmax + 1); }
inline ptrdiff_t full_mind(int lmax, int m) {     // start of block m in the all-m layout (comm_map_mod.f90:228-261, P = 1)
    return m == 0 ? 0 : 2 * ((ptrdiff_t)m * (lmax + 1) - (ptrdiff_t)m * (m - 1) / 2) - (lmax + 1);
}
// local (this rank's m blocks, in ms order) <-> full (all m) packed a_lm
void local_to_full(const sharp_alm_info* a, const double* loc, double* full) {
    ptrdiff_t o = 0;
    for (int m : a->ms) {
        const ptrdiff_t n = m == 0 ? a->lmax + 1 : 2 * (ptrdiff_t)(a->lmax + 1 - m);
        std::copy(loc + o, loc + o + n, full + full_mind(a->lmax, m));
        o += n;
    }
}
void full_to_local(const sharp_alm_info* a, const double* full, double* loc) {
    ptrdiff_t o = 0;
    for (int m : a->ms) {
        const ptrdiff_t n = m == 0 ? a->lmax + 1 : 2 * (ptrdiff_t)(a->lmax + 1 - m);
        std::copy(full + full_mind(a->lmax, m), full + full_mind(a->lmax, m) + n, loc + o);
        o += n;
    }
}
}  // namespace

extern "C" {

void sharp_make_mmajor_real_packed_alm_info(int lmax, int stride, int nm, const int* ms, sharp_alm_info** out) {
    if (stride != 1) die("alm stride must be 1 (commander3/src/sharp.f90:128)");
    auto* a = new sharp_alm_info;
    a->lmax = lmax;
    a->count = 0;
    a->all_m = nm == lmax + 1;
    for (int i = 0; i < nm; ++i) {
        const int m = ms ? ms[i] : i;
        if (m < 0 || m > lmax) die("m outside 0..lmax");
        if (m != i) a->all_m = false;
        a->ms.push_back(m);
        a->count += (m == 0) ? (lmax + 1) : 2 * (lmax + 1 - m);
    }
    *out = a;
}

void sharp_make_general_alm_info(int lmax, int nm, int stride, const int* mval, const ptrdiff_t* mvstart, int flags,
                                 sharp_alm_info** out) {
    (void)mvstart;
    if (!(flags & 1)) die("sharp_make_general_alm_info: only SHARP_PACKED real m-major layouts are supported");
    sharp_make_mmajor_real_packed_alm_info(lmax, stride, nm, mval, out);
}

ptrdiff_t sharp_alm_count(const sharp_alm_info* self) { return self->count; }
void sharp_destroy_alm_info(sharp_alm_info* info) { delete info; }

void sharp_make_subset_healpix_geom_info(int nside, int stride, int nrings, const int* rings, const double* weight,
                                         sharp_geom_info** out) {
    if (stride != 1) die("map stride must be 1 (commander3/src/sharp.f90:158)");
    auto* g = new sharp_geom_info;
    g->nside = nside;
    g->npix = 0;
    std::vector<char> have(4 * nside, 0);
    if (rings)
        for (int i = 0; i < nrings; ++i) {
            if (rings[i] < 1 || rings[i] > 4 * nside - 1) die("ring number outside 1..4*nside-1");
            have[rings[i]] = 1;
        }
    else for (int r = 1; r <= 4 * nside - 1; ++r) have[r] = 1;
    for (int r = 1; r <= 2 * nside; ++r) {
        if (!have[r]) {
            if (r < 2 * nside && have[4 * nside - r]) die("ring subset must contain north/south mirror pairs");
            continue;
        }
        if (r < 2 * nside && !have[4 * nside - r]) die("ring subset must contain north/south mirror pairs");
        g->north.push_back(r);
        const ptrdiff_t nphi = r < nside ? 4 * r : 4 * nside;
        g->npix += (r == 2 * nside) ? nphi : 2 * nphi;
    }
    if (weight) g->weight.assign(weight, weight + 2 * nside);
    *out = g;
}

ptrdiff_t sharp_map_size(const sharp_geom_info* info) { return info->npix; }

void sharp_destroy_geom_info(sharp_geom_info* info) {
    if (!info) return;
    for (auto& kv : info->plans) cmdr_sht_plan_destroy(kv.second);
    delete info;
}

static cmdr_sht_plan* plan_of(const sharp_geom_info* g, int lmax, int spin) {
    const int key = 2 * lmax + (spin == 2 ? 1 : 0);
    auto it = g->plans.find(key);
    if (it == g->plans.end()) {
        cmdr_sht_plan* p = nullptr;
        const int rc = spin == 2 ? cmdr_sht_plan_create_pol(g->nside, lmax, (int)g->north.size(), g->north.data(),
                                                            g->weight.empty() ? nullptr : g->weight.data(), 2, &p)
                                 : cmdr_sht_plan_create(g->nside, lmax, (int)g->north.size(), g->north.data(),
                                                        g->weight.empty() ? nullptr : g->weight.data(), 1, &p);
        if (rc != 0) die(cmdr_last_error());
        it = g->plans.emplace(key, p).first;
    }
    return it->second;
}

static void check_job(int type, int spin, int flags) {
    if (spin != 0 && spin != 2) die("only spin 0 and spin 2 are supported (what Commander issues)");
    if (!(flags & SHARP_DP)) die("single precision is not supported (Commander always sets SHARP_DP)");
    if (flags & SHARP_ADD) die("SHARP_ADD is not supported (never set by Commander)");
    if (type < SHARP_YtW || type > SHARP_WY) die("unsupported job type");
}

static void run_plan(cmdr_sht_plan* p, int type, int spin, double* const* pa, double* const* pm) {
    if (spin == 2) {   // alm = (E, B), map = (Q, U): sharp.f90:203-224 with nmaps = 2
        if (cmdr_sht_execute_spin2(p, type, pa[0], pa[1], pm[0], pm[1]) != 0) die(cmdr_last_error());
    } else if (cmdr_sht_execute(p, type, 1, pa, pm) != 0) {
        die(cmdr_last_error());
    }
}

void sharp_execute(int type, int spin, void* alm, void* map, const sharp_geom_info* g, const sharp_alm_info* a,
                   int flags, double* time, unsigned long long* opcnt) {
    check_job(type, spin, flags);
    if (!a->all_m) die("sharp_execute: every m must be local (ms = 0..lmax); distributed a_lm go through sharp_execute_mpi");
    const auto t0 = std::chrono::steady_clock::now();
    run_plan(plan_of(g, a->lmax, spin), type, spin, static_cast<double* const*>(alm), static_cast<double* const*>(map));
    if (time) *time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (opcnt) *opcnt = 0;
}

void cmdr_sharp_register_comm(int comm, cmdr_sharp_allreduce_fn fn, void* user) {
    if (fn) comms()[comm] = CommEntry{fn, user};
    else comms().erase(comm);
}

// libsharp2 redistributes a_lm (by m) and rings between the ranks of comm inside this call (sharp.f90:96-104).  Here
// every rank transforms ITS rings with ALL m on its GPU, so the exchange is one sum over the communicator of the full
// packed a_lm: before the transform for Y / WY (each rank contributes the m blocks it owns), after it for Yt / YtW
// (each rank contributes its rings' partial sums, then keeps the m blocks it owns).
void sharp_execute_mpi_fortran(int comm, int type, int spin, void* alm, void* map, const sharp_geom_info* g,
                               const sharp_alm_info* a, int flags, double* time, unsigned long long* opcnt) {
    auto it = comms().find(comm);
    if (it == comms().end()) {   // unregistered communicator: must be a one-rank chain (every m and every ring local)
        sharp_execute(type, spin, alm, map, g, a, flags, time, opcnt);
        return;
    }
    check_job(type, spin, flags);
    const auto t0 = std::chrono::steady_clock::now();
    const int ncol = spin == 2 ? 2 : 1;
    const ptrdiff_t nf = full_count(a->lmax);
    double* const* pa = static_cast<double* const*>(alm);
    std::vector<double> full((size_t)nf * ncol, 0.0);
    double* cols[2] = {full.data(), full.data() + nf};
    const bool synth = type == SHARP_Y || type == SHARP_WY;
    if (synth) {
        for (int k = 0; k < ncol; ++k) if (a->count > 0) local_to_full(a, pa[k], cols[k]);
        it->second.fn(it->second.user, full.data(), (int64_t)full.size());
    }
    if (g->npix > 0) {
        run_plan(plan_of(g, a->lmax, spin), type, spin, cols, static_cast<double* const*>(map));
    } else if (!synth) {
        std::fill(full.begin(), full.end(), 0.0);      // a rank without rings contributes nothing
    }
    if (!synth) {
        it->second.fn(it->second.user, full.data(), (int64_t)full.size());
        for (int k = 0; k < ncol; ++k) if (a->count > 0) full_to_local(a, cols[k], pa[k]);
    }
    if (time) *time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (opcnt) *opcnt = 0;
}

}  // extern "C"
