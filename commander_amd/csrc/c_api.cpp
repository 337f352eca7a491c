// extern "C" surface of libcmdr_hip.so (declared in include/cmdr_hip.h).
#include "../../include/cmdr_hip.h"

#include <cstring>
#include <string>

#include "common.hpp"
#include "sht_plan.hpp"

namespace {
thread_local std::string g_err;

template <typename F>
int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    } catch (...) {
        g_err = "unknown error";
        return -2;
    }
}
}  // namespace

struct cmdr_sht_plan {
    std::unique_ptr<cmdr::ShtPlan> p;
    cmdr::DevBuf<double> alm, map;  // staging for the host-pointer entry point
};

extern "C" {

const char* cmdr_last_error(void) { return g_err.c_str(); }

int cmdr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cmdr_set_device(int device) {
    return guarded([&] { CMDR_HIP_CHECK(hipSetDevice(device)); });
}

int cmdr_device_synchronize(void) {
    return guarded([&] { CMDR_HIP_CHECK(hipDeviceSynchronize()); });
}

int cmdr_dev_alloc(size_t nbytes, void** out) {
    return guarded([&] { CMDR_HIP_CHECK(hipMalloc(out, nbytes)); });
}
int cmdr_dev_free(void* p) {
    return guarded([&] { CMDR_HIP_CHECK(hipFree(p)); });
}
int cmdr_memcpy_h2d(void* dst, const void* src, size_t n) {
    return guarded([&] { CMDR_HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); });
}
int cmdr_memcpy_d2h(void* dst, const void* src, size_t n) {
    return guarded([&] { CMDR_HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); });
}

int cmdr_sht_plan_create(int nside, int lmax, int nrings, const int* rings, const double* wring, int max_maps,
                         cmdr_sht_plan** out) {
    return guarded([&] {
        CMDR_REQUIRE(out != nullptr, "out is NULL");
        CMDR_REQUIRE(cmdr_device_count() > 0, "no HIP device available: libcmdr_hip has no CPU path");
        std::vector<int> r;
        if (rings && nrings > 0) r.assign(rings, rings + nrings);
        auto* h = new cmdr_sht_plan;
        try {
            h->p = std::make_unique<cmdr::ShtPlan>(nside, lmax, r, wring, max_maps);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
    });
}

int cmdr_sht_plan_destroy(cmdr_sht_plan* plan) {
    return guarded([&] { delete plan; });
}

int64_t cmdr_sht_nalm(const cmdr_sht_plan* plan) { return plan ? plan->p->nalm() : -1; }
int64_t cmdr_sht_npix(const cmdr_sht_plan* plan) { return plan ? plan->p->npix_local() : -1; }

int cmdr_sht_execute_dev(cmdr_sht_plan* plan, int job, int nmaps, double* alm_dev, int64_t alm_stride,
                         double* map_dev, int64_t map_stride) {
    return guarded([&] {
        CMDR_REQUIRE(plan != nullptr, "plan is NULL");
        CMDR_REQUIRE(nmaps >= 1, "nmaps must be >= 1");
        cmdr::ShtPlan& P = *plan->p;
        switch (job) {
            case CMDR_Y: P.alm2map(alm_dev, alm_stride, map_dev, map_stride, nmaps, false, nullptr); break;
            case CMDR_WY: P.alm2map(alm_dev, alm_stride, map_dev, map_stride, nmaps, true, nullptr); break;
            case CMDR_Yt: P.map2alm(map_dev, map_stride, alm_dev, alm_stride, nmaps, false, nullptr); break;
            case CMDR_YtW: P.map2alm(map_dev, map_stride, alm_dev, alm_stride, nmaps, true, nullptr); break;
            default: throw cmdr::Error("unknown SHT job type " + std::to_string(job));
        }
        CMDR_HIP_CHECK(hipGetLastError());
        CMDR_HIP_CHECK(hipStreamSynchronize(nullptr));
    });
}

int cmdr_sht_execute(cmdr_sht_plan* plan, int job, int nmaps, double* const* alm, double* const* map) {
    return guarded([&] {
        CMDR_REQUIRE(plan != nullptr, "plan is NULL");
        CMDR_REQUIRE(nmaps >= 1 && alm && map, "bad arguments");
        cmdr::ShtPlan& P = *plan->p;
        const int64_t na = P.nalm(), np = P.npix_local();
        plan->alm.ensure((size_t)na * nmaps);
        plan->map.ensure((size_t)np * nmaps);
        const bool synth = (job == CMDR_Y || job == CMDR_WY);
        for (int k = 0; k < nmaps; ++k) {
            if (synth)
                CMDR_HIP_CHECK(hipMemcpy(plan->alm.get() + k * na, alm[k], na * sizeof(double), hipMemcpyHostToDevice));
            else
                CMDR_HIP_CHECK(hipMemcpy(plan->map.get() + k * np, map[k], np * sizeof(double), hipMemcpyHostToDevice));
        }
        if (cmdr_sht_execute_dev(plan, job, nmaps, plan->alm.get(), na, plan->map.get(), np) != 0)
            throw cmdr::Error(g_err);
        for (int k = 0; k < nmaps; ++k) {
            if (synth)
                CMDR_HIP_CHECK(hipMemcpy(map[k], plan->map.get() + k * np, np * sizeof(double), hipMemcpyDeviceToHost));
            else
                CMDR_HIP_CHECK(hipMemcpy(alm[k], plan->alm.get() + k * na, na * sizeof(double), hipMemcpyDeviceToHost));
        }
    });
}

}  // extern "C"
