// The literal libsharp2 entry points of include/cmdr_sharp.h, implemented on top of the SHT-level C ABI.
#include "../../include/cmdr_sharp.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../include/cmdr_hip.h"

struct sharp_alm_info {
    int lmax;
    std::vector<int> ms;
    ptrdiff_t count;
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
}  // namespace

extern "C" {

void sharp_make_mmajor_real_packed_alm_info(int lmax, int stride, int nm, const int* ms, sharp_alm_info** out) {
    if (stride != 1) die("alm stride must be 1 (commander3/src/sharp.f90:128)");
    auto* a = new sharp_alm_info;
    a->lmax = lmax;
    a->count = 0;
    for (int i = 0; i < nm; ++i) {
        const int m = ms ? ms[i] : i;
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
    if (rings) for (int i = 0; i < nrings; ++i) have[rings[i]] = 1;
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

void sharp_execute(int type, int spin, void* alm, void* map, const sharp_geom_info* g, const sharp_alm_info* a,
                   int flags, double* time, unsigned long long* opcnt) {
    if (spin != 0 && spin != 2) die("only spin 0 and spin 2 are supported (what Commander issues)");
    if (!(flags & SHARP_DP)) die("single precision is not supported (Commander always sets SHARP_DP)");
    if (flags & SHARP_ADD) die("SHARP_ADD is not supported (never set by Commander)");
    if (type < SHARP_YtW || type > SHARP_WY) die("unsupported job type");
    if ((int)a->ms.size() != a->lmax + 1) die("every m must be local: use one MPI rank per chain/GPU");
    for (int i = 0; i <= a->lmax; ++i) if (a->ms[i] != i) die("ms must be 0..lmax in order");
    const int key = 2 * a->lmax + (spin == 2 ? 1 : 0);
    auto it = g->plans.find(key);
    if (it == g->plans.end()) {
        cmdr_sht_plan* p = nullptr;
        const int rc = spin == 2 ? cmdr_sht_plan_create_pol(g->nside, a->lmax, (int)g->north.size(), g->north.data(),
                                                            g->weight.empty() ? nullptr : g->weight.data(), 2, &p)
                                 : cmdr_sht_plan_create(g->nside, a->lmax, (int)g->north.size(), g->north.data(),
                                                        g->weight.empty() ? nullptr : g->weight.data(), 1, &p);
        if (rc != 0) die(cmdr_last_error());
        it = g->plans.emplace(key, p).first;
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (spin == 2) {   // alm = (E, B), map = (Q, U): sharp.f90:203-224 with nmaps = 2
        double* const* pa = static_cast<double* const*>(alm);
        double* const* pm = static_cast<double* const*>(map);
        if (cmdr_sht_execute_spin2(it->second, type, pa[0], pa[1], pm[0], pm[1]) != 0) die(cmdr_last_error());
    } else if (cmdr_sht_execute(it->second, type, 1, static_cast<double* const*>(alm),
                                static_cast<double* const*>(map)) != 0) {
        die(cmdr_last_error());
    }
    if (time) *time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (opcnt) *opcnt = 0;
}

void sharp_execute_mpi_fortran(int comm, int type, int spin, void* alm, void* map, const sharp_geom_info* g,
                               const sharp_alm_info* a, int flags, double* time, unsigned long long* opcnt) {
    (void)comm;  // a one-rank communicator: nothing to exchange (checked through the "every m local" rule)
    sharp_execute(type, spin, alm, map, g, a, flags, time, opcnt);
}

}  // extern "C"
