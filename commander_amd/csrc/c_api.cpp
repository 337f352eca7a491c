// extern "C" surface of libcmdr_hip.so (declared in include/cmdr_hip.h).
#include "../../include/cmdr_hip.h"

#include <cstring>
#include <string>

#include "cl_sampler.hpp"
#include "common.hpp"
#include "cr_system.hpp"
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

namespace cmdr {
void set_last_error(const char* msg) { g_err = msg ? msg : ""; }   // for the entry points that live in other files
}

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
int cmdr_dev_mem_info(size_t* free_bytes, size_t* total_bytes) {
    return guarded([&] {
        CMDR_REQUIRE(free_bytes && total_bytes, "NULL argument");
        CMDR_HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes));
    });
}
int cmdr_memcpy_h2d(void* dst, const void* src, size_t n) {
    return guarded([&] { CMDR_HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice)); });
}
int cmdr_memcpy_d2h(void* dst, const void* src, size_t n) {
    return guarded([&] { CMDR_HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost)); });
}
int cmdr_host_register(void* p, size_t n) {
    return guarded([&] {
        CMDR_REQUIRE(p != nullptr && n > 0, "bad arguments");
        CMDR_HIP_CHECK(hipHostRegister(p, n, hipHostRegisterDefault));
    });
}
int cmdr_host_unregister(void* p) {
    return guarded([&] {
        CMDR_REQUIRE(p != nullptr, "NULL argument");
        CMDR_HIP_CHECK(hipHostUnregister(p));
    });
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

int cmdr_sht_plan_create_pol(int nside, int lmax, int nrings, const int* rings, const double* wring, int max_maps,
                             cmdr_sht_plan** out) {
    return guarded([&] {
        CMDR_REQUIRE(out != nullptr, "out is NULL");
        CMDR_REQUIRE(cmdr_device_count() > 0, "no HIP device available: libcmdr_hip has no CPU path");
        std::vector<int> r;
        if (rings && nrings > 0) r.assign(rings, rings + nrings);
        auto* h = new cmdr_sht_plan;
        try {
            h->p = std::make_unique<cmdr::ShtPlan>(nside, lmax, r, wring, std::max(2, max_maps), true);
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
    });
}

int cmdr_sht_execute_spin2_dev(cmdr_sht_plan* plan, int job, double* almE_dev, double* almB_dev, double* mapQ_dev,
                               double* mapU_dev) {
    return guarded([&] {
        CMDR_REQUIRE(plan && almE_dev && almB_dev && mapQ_dev && mapU_dev, "bad arguments");
        cmdr::ShtPlan& P = *plan->p;
        switch (job) {
            case CMDR_Y: P.alm2map_spin2(almE_dev, almB_dev, mapQ_dev, mapU_dev, false, nullptr); break;
            case CMDR_WY: P.alm2map_spin2(almE_dev, almB_dev, mapQ_dev, mapU_dev, true, nullptr); break;
            case CMDR_Yt: P.map2alm_spin2(mapQ_dev, mapU_dev, almE_dev, almB_dev, false, nullptr); break;
            case CMDR_YtW: P.map2alm_spin2(mapQ_dev, mapU_dev, almE_dev, almB_dev, true, nullptr); break;
            default: throw cmdr::Error("unknown SHT job type " + std::to_string(job));
        }
        CMDR_HIP_CHECK(hipGetLastError());
        CMDR_HIP_CHECK(hipStreamSynchronize(nullptr));
    });
}

int cmdr_sht_execute_spin2(cmdr_sht_plan* plan, int job, double* almE, double* almB, double* mapQ, double* mapU) {
    return guarded([&] {
        CMDR_REQUIRE(plan && almE && almB && mapQ && mapU, "bad arguments");
        cmdr::ShtPlan& P = *plan->p;
        const int64_t na = P.nalm(), np = P.npix_local();
        plan->alm.ensure((size_t)na * 2);
        plan->map.ensure((size_t)np * 2);
        const bool synth = (job == CMDR_Y || job == CMDR_WY);
        double *dE = plan->alm.get(), *dB = dE + na, *dQ = plan->map.get(), *dU = dQ + np;
        if (synth) {
            CMDR_HIP_CHECK(hipMemcpy(dE, almE, na * sizeof(double), hipMemcpyHostToDevice));
            CMDR_HIP_CHECK(hipMemcpy(dB, almB, na * sizeof(double), hipMemcpyHostToDevice));
        } else {
            CMDR_HIP_CHECK(hipMemcpy(dQ, mapQ, np * sizeof(double), hipMemcpyHostToDevice));
            CMDR_HIP_CHECK(hipMemcpy(dU, mapU, np * sizeof(double), hipMemcpyHostToDevice));
        }
        if (cmdr_sht_execute_spin2_dev(plan, job, dE, dB, dQ, dU) != 0) throw cmdr::Error(g_err);
        if (synth) {
            CMDR_HIP_CHECK(hipMemcpy(mapQ, dQ, np * sizeof(double), hipMemcpyDeviceToHost));
            CMDR_HIP_CHECK(hipMemcpy(mapU, dU, np * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            CMDR_HIP_CHECK(hipMemcpy(almE, dE, na * sizeof(double), hipMemcpyDeviceToHost));
            CMDR_HIP_CHECK(hipMemcpy(almB, dB, na * sizeof(double), hipMemcpyDeviceToHost));
        }
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

// ------------------------------------------------------------------------------------------------- CR level
struct cmdr_ctx {
    std::unique_ptr<cmdr::CrSystem> sys;
    cmdr::DevBuf<double> hx, hy, hz;           // staging for the host-pointer entry points
    std::vector<cmdr::DevBuf<double>> hmaps;   // staged band maps (resid, xi) of the last cmdr_compute_rhs: the 'chisq'
                                               // criterion of a later cmdr_solve reads the residuals, so nothing else
                                               // may be staged here
    std::vector<cmdr::DevBuf<double>> hres;    // staging of cmdr_compute_residual (data, resid)
};

extern "C" {

int cmdr_ctx_create(int device, cmdr_ctx** out) {
    return guarded([&] {
        CMDR_REQUIRE(out != nullptr, "out is NULL");
        CMDR_REQUIRE(cmdr_device_count() > device && device >= 0,
                     "no such HIP device: libcmdr_hip has no CPU path");
        auto* c = new cmdr_ctx;
        try {
            c->sys = std::make_unique<cmdr::CrSystem>(device);
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
    });
}
int cmdr_ctx_destroy(cmdr_ctx* ctx) {
    return guarded([&] { delete ctx; });
}
int cmdr_ctx_set_rings(cmdr_ctx* ctx, int nside, int nrings, const int* rings) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && rings && nrings > 0, "bad arguments");
        ctx->sys->set_rings(nside, std::vector<int>(rings, rings + nrings));
    });
}
int cmdr_ctx_set_allreduce(cmdr_ctx* ctx, cmdr_allreduce_fn fn, void* user) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_allreduce(fn, user);
    });
}
int cmdr_ctx_set_allreduce_stream(cmdr_ctx* ctx, cmdr_allreduce_stream_fn fn, void* user) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_allreduce_stream(fn, user);
    });
}
int cmdr_ctx_set_band_sharding(cmdr_ctx* ctx, cmdr_allreduce_fn rings_fn, void* user, int ring_replicas) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && ring_replicas >= 1, "bad arguments");
        ctx->sys->set_band_sharding(rings_fn, user, ring_replicas);
    });
}
int cmdr_rccl_unique_id(char* out128) {
    return guarded([&] {
        CMDR_REQUIRE(out128, "out is NULL");
        cmdr::RcclComm::unique_id(out128);
    });
}
int cmdr_rccl_version(void) {
    int v = -1;
    (void)guarded([&] { v = cmdr::RcclComm::version(); });
    return v;
}
int cmdr_ctx_init_rccl(cmdr_ctx* ctx, const char* id128, int rank, int nranks) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && id128, "bad arguments");
        ctx->sys->init_rccl(id128, rank, nranks);
    });
}
int cmdr_ctx_rccl_split_rings(cmdr_ctx* ctx, int band_group, int ring_index, int ring_replicas) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->rccl_split_rings(band_group, ring_index, ring_replicas);
    });
}
int cmdr_ctx_set_vector_slicing(cmdr_ctx* ctx, int rank, int nranks) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_vector_slicing(rank, nranks);
    });
}
int cmdr_ctx_drop_rccl(cmdr_ctx* ctx) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->drop_rccl();
    });
}
int cmdr_ctx_rccl_size(cmdr_ctx* ctx) {
    int n = -1;
    (void)guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        n = ctx->sys->rccl_size();
    });
    return n;
}
int cmdr_ctx_set_only_pol(cmdr_ctx* ctx, int only_pol) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_only_pol(only_pol != 0);
    });
}
int cmdr_ctx_set_literal_quirks(cmdr_ctx* ctx, int on) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_literal_quirks(on != 0);
    });
}
int cmdr_band_add(cmdr_ctx* ctx, int nside, int lmax, int nmaps, const double* siN, const double* b_l,
                  double mb_eff, const double* sg_mask, const double* wring) {
    int idx = -1;
    const int rc = guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        idx = ctx->sys->add_band(nside, lmax, nmaps, siN, b_l, mb_eff, sg_mask, wring);
    });
    return rc == 0 ? idx : rc;
}
int cmdr_comp_add(cmdr_ctx* ctx, int lmax_amp, int nmaps, int lmax_cl, const double* sqrtS_mat,
                  const double* sqrtInvS_mat, const double* S_mat, const double* F_mean, int active) {
    int idx = -1;
    const int rc = guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        idx = ctx->sys->add_comp(lmax_amp, nmaps, lmax_cl, sqrtS_mat, sqrtInvS_mat, S_mat, F_mean, active);
    });
    return rc == 0 ? idx : rc;
}
int cmdr_finalize(cmdr_ctx* ctx) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->finalize();
    });
}
int64_t cmdr_ncr(const cmdr_ctx* ctx) { return ctx ? ctx->sys->ncr() : -1; }
int64_t cmdr_band_npix(const cmdr_ctx* ctx, int band) {
    if (!ctx || band < 0 || band >= ctx->sys->nband()) return -1;
    return ctx->sys->band_npix(band);
}
int cmdr_precond_init_diag(cmdr_ctx* ctx) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->precond_init_diag();
    });
}
int cmdr_precond_update_diag(cmdr_ctx* ctx) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->precond_update_diag();
        ctx->sys->sync();
    });
}
int cmdr_precond_set_lowl(cmdr_ctx* ctx, int comp, int lmax_pre_lowl, const int* nside_lowres,
                          const double* const* siN_lowres) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_lowl(comp, lmax_pre_lowl, nside_lowres, siN_lowres);
    });
}
int cmdr_precond_init_pseudoinv(cmdr_ctx* ctx) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->precond_init_pseudoinv();
    });
}
int cmdr_precond_update_pseudoinv(cmdr_ctx* ctx) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->precond_update_pseudoinv();
        ctx->sys->sync();
    });
}
int cmdr_get_alpha_nu(cmdr_ctx* ctx, int band, double* out_host) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && out_host && band >= 0 && band < ctx->sys->nband(), "bad arguments");
        const std::vector<double>& a = ctx->sys->alpha_nu(band);
        CMDR_REQUIRE(!a.empty(), "cmdr_precond_init_pseudoinv has not run");
        std::copy(a.begin(), a.end(), out_host);
    });
}
int cmdr_band_set_qucov(cmdr_ctx* ctx, int band, const double* iN, const double* siN_mat) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_band_qucov(band, iN, siN_mat);
    });
}
int cmdr_compact_add(cmdr_ctx* ctx, int nparam, const double* sigma, const double* mean, int active) {
    int idx = -1;
    const int rc = guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        idx = ctx->sys->add_compact(nparam, sigma, mean, active);
    });
    return rc == 0 ? idx : rc;
}
int cmdr_compact_set_band(cmdr_ctx* ctx, int block, int band, int64_t nnz, const int64_t* cell, const int* param,
                          const double* val) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_compact_band(block, band, nnz, cell, param, val);
    });
}
int cmdr_comp_set_mixing_map(cmdr_ctx* ctx, int comp, int band, const double* F, int nmaps) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_mixing_map(comp, band, F, nmaps);
    });
}
int cmdr_comp_set_cl_diag(cmdr_ctx* ctx, int comp, const double* cl) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_cl_diag(comp, cl);
    });
}
int cmdr_comp_set_cl(cmdr_ctx* ctx, int comp, const double* sqrtS_mat, const double* sqrtInvS_mat, const double* S_mat) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_comp_cl(comp, sqrtS_mat, sqrtInvS_mat, S_mat);
    });
}
int cmdr_comp_set_f_mean(cmdr_ctx* ctx, int comp, const double* F_mean) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_comp_f_mean(comp, F_mean);
    });
}
int cmdr_comp_set_active(cmdr_ctx* ctx, int comp, int active) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_active(0, comp, active);
    });
}
int cmdr_compact_set_active(cmdr_ctx* ctx, int block, int active) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_active(1, block, active);
    });
}
int cmdr_cl_update_S(int lmax, int nmaps, int lmin, const double* Dl, const double* RJ2unit, double* sqrtS_mat,
                     double* sqrtInvS_mat, double* S_mat) {
    int n = 0;
    const int rc = guarded([&] { n = cmdr::cl_update_S(lmax, nmaps, lmin, Dl, RJ2unit, sqrtS_mat, sqrtInvS_mat, S_mat); });
    return rc < 0 ? rc : n;
}
int cmdr_cl_sample_lookup(int lmax, int lmin_lookup, int lmax_lookup, int nmodel, const double* Dl_lookup,
                          const int* active, const double* sigma_l, const double* S_mat, const double* RJ2unit,
                          double uniform, double* Dl, int* chosen) {
    int r = 0;
    const int rc = guarded([&] {
        r = cmdr::cl_sample_lookup(lmax, lmin_lookup, lmax_lookup, nmodel, Dl_lookup, active, sigma_l, S_mat, RJ2unit,
                                   uniform, Dl, chosen);
    });
    return rc < 0 ? rc : r;
}
double cmdr_cl_apod(int l, int l_apod, int lmax, int lmax_prior, int positive) {
    return cmdr::cl_apod(l, l_apod, lmax, lmax_prior, positive != 0);
}
int cmdr_cl_apply_apod(int lmax, int nmaps, int l_apod, int lmax_prior, double* sqrtS_mat, double* sqrtInvS_mat,
                       double* S_mat) {
    return guarded([&] { cmdr::cl_apply_apod(lmax, nmaps, l_apod, lmax_prior, sqrtS_mat, sqrtInvS_mat, S_mat); });
}
int cmdr_cl_sample_binned(int lmax, int nmaps, const double* sigma_l, const double* S_mat, const double* RJ2unit, int nbin,
                          const cmdr_cl_bin* bins, const double* uniform, int nuniform, double* Dl, int* nused) {
    int r = 0;
    const int rc = guarded([&] {
        r = cmdr::cl_sample_binned(lmax, nmaps, sigma_l, S_mat, RJ2unit, nbin, bins, uniform, nuniform, Dl, nused);
    });
    return rc < 0 ? rc : r;
}
int cmdr_get_invN_diag(cmdr_ctx* ctx, int band, double* out_host) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && out_host && band >= 0 && band < ctx->sys->nband(), "bad arguments");
        const double* p = ctx->sys->invN_diag_dev(band);
        CMDR_REQUIRE(p != nullptr, "cmdr_precond_init_diag has not run");
        // nmaps columns of nalm
        const int64_t n = ctx->sys->band_nalm(band);
        CMDR_HIP_CHECK(hipMemcpy(out_host, p, sizeof(double) * n, hipMemcpyDeviceToHost));
    });
}

int cmdr_matmulA_dev(cmdr_ctx* ctx, const double* x, double* y) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && x && y, "bad arguments");
        ctx->sys->matmulA(x, y);
        CMDR_HIP_CHECK(hipGetLastError());
        ctx->sys->sync();
    });
}
int cmdr_invM_dev(cmdr_ctx* ctx, const double* x, double* y) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && x && y, "bad arguments");
        ctx->sys->invM(x, y);
        CMDR_HIP_CHECK(hipGetLastError());
        ctx->sys->sync();
    });
}
int cmdr_matmulA(cmdr_ctx* ctx, const double* x, double* y) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && x && y, "bad arguments");
        const size_t n = (size_t)ctx->sys->ncr();
        ctx->hx.ensure(n);
        ctx->hy.ensure(n);
        CMDR_HIP_CHECK(hipMemcpy(ctx->hx.get(), x, n * sizeof(double), hipMemcpyHostToDevice));
        if (cmdr_matmulA_dev(ctx, ctx->hx.get(), ctx->hy.get()) != 0) throw cmdr::Error(g_err);
        CMDR_HIP_CHECK(hipMemcpy(y, ctx->hy.get(), n * sizeof(double), hipMemcpyDeviceToHost));
    });
}
int cmdr_invM(cmdr_ctx* ctx, const double* x, double* y) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && x && y, "bad arguments");
        const size_t n = (size_t)ctx->sys->ncr();
        ctx->hx.ensure(n);
        ctx->hy.ensure(n);
        CMDR_HIP_CHECK(hipMemcpy(ctx->hx.get(), x, n * sizeof(double), hipMemcpyHostToDevice));
        if (cmdr_invM_dev(ctx, ctx->hx.get(), ctx->hy.get()) != 0) throw cmdr::Error(g_err);
        CMDR_HIP_CHECK(hipMemcpy(y, ctx->hy.get(), n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int cmdr_compute_rhs_dev(cmdr_ctx* ctx, int sample, const double* const* resid, const double* const* xi,
                         const double* eta, const double* mu, double* rhs) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && resid && rhs, "bad arguments");
        CMDR_REQUIRE(!sample || (xi && eta), "operation 'sample' needs xi and eta");
        ctx->sys->compute_rhs(sample != 0, resid, xi, eta, mu, rhs);
        CMDR_HIP_CHECK(hipGetLastError());
        ctx->sys->sync();
    });
}
int cmdr_compute_rhs(cmdr_ctx* ctx, int sample, const double* const* resid, const double* const* xi,
                     const double* eta, const double* mu, double* rhs) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && resid && rhs, "bad arguments");
        CMDR_REQUIRE(!sample || (xi && eta), "operation 'sample' needs xi and eta");
        const int nb = ctx->sys->nband();
        const size_t n = (size_t)ctx->sys->ncr();
        ctx->hmaps.resize(2 * (size_t)nb);
        std::vector<const double*> dres(nb), dxi(nb, nullptr);
        for (int b = 0; b < nb; ++b) {
            const size_t np = (size_t)ctx->sys->band_npix(b) * 1;  // per column; nmaps columns are contiguous
            // nmaps is folded into the staging size through siN's own size on the device side
            const size_t tot = np * (size_t)(ctx->sys->band_nmaps(b));
            ctx->hmaps[2 * b].ensure(tot);
            CMDR_HIP_CHECK(hipMemcpy(ctx->hmaps[2 * b].get(), resid[b], tot * sizeof(double), hipMemcpyHostToDevice));
            dres[b] = ctx->hmaps[2 * b].get();
            if (sample) {
                ctx->hmaps[2 * b + 1].ensure(tot);
                CMDR_HIP_CHECK(hipMemcpy(ctx->hmaps[2 * b + 1].get(), xi[b], tot * sizeof(double), hipMemcpyHostToDevice));
                dxi[b] = ctx->hmaps[2 * b + 1].get();
            }
        }
        ctx->hx.ensure(n);
        ctx->hy.ensure(n);
        ctx->hz.ensure(n);
        if (sample) CMDR_HIP_CHECK(hipMemcpy(ctx->hx.get(), eta, n * sizeof(double), hipMemcpyHostToDevice));
        if (mu) CMDR_HIP_CHECK(hipMemcpy(ctx->hz.get(), mu, n * sizeof(double), hipMemcpyHostToDevice));
        if (cmdr_compute_rhs_dev(ctx, sample, dres.data(), dxi.data(), sample ? ctx->hx.get() : nullptr,
                                 mu ? ctx->hz.get() : nullptr, ctx->hy.get()) != 0)
            throw cmdr::Error(g_err);
        CMDR_HIP_CHECK(hipMemcpy(rhs, ctx->hy.get(), n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int cmdr_compute_residual_dev(cmdr_ctx* ctx, const double* amp, const double* const* data, double* const* resid) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && amp && data && resid, "bad arguments");
        ctx->sys->compute_residual(amp, data, resid);
        CMDR_HIP_CHECK(hipGetLastError());
        ctx->sys->sync();
    });
}
int cmdr_compute_residual(cmdr_ctx* ctx, const double* amp, const double* const* data, double* const* resid) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && amp && data && resid, "bad arguments");
        const int nb = ctx->sys->nband();
        const size_t n = (size_t)ctx->sys->ncr();
        ctx->hres.resize(2 * (size_t)nb);
        std::vector<const double*> dd(nb);
        std::vector<double*> dr(nb);
        std::vector<size_t> tot(nb);
        for (int b = 0; b < nb; ++b) {
            tot[b] = (size_t)ctx->sys->band_npix(b) * (size_t)ctx->sys->band_nmaps(b);
            ctx->hres[2 * b].ensure(tot[b]);
            ctx->hres[2 * b + 1].ensure(tot[b]);
            CMDR_HIP_CHECK(hipMemcpy(ctx->hres[2 * b].get(), data[b], tot[b] * sizeof(double), hipMemcpyHostToDevice));
            dd[b] = ctx->hres[2 * b].get();
            dr[b] = ctx->hres[2 * b + 1].get();
        }
        ctx->hx.ensure(n);
        CMDR_HIP_CHECK(hipMemcpy(ctx->hx.get(), amp, n * sizeof(double), hipMemcpyHostToDevice));
        if (cmdr_compute_residual_dev(ctx, ctx->hx.get(), dd.data(), dr.data()) != 0) throw cmdr::Error(g_err);
        for (int b = 0; b < nb; ++b)
            CMDR_HIP_CHECK(hipMemcpy(resid[b], dr[b], tot[b] * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int cmdr_apply_mono_dipole_prior_dev(cmdr_ctx* ctx, int comp, double* amp, int nside, const double* b_l_out,
                                     const double* mask, int type, double* mu) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && amp && mask && mu, "bad arguments");
        ctx->sys->apply_mono_dipole_prior(comp, amp, nside, b_l_out, mask, type, mu);
        CMDR_HIP_CHECK(hipGetLastError());
    });
}
int cmdr_apply_mono_dipole_prior(cmdr_ctx* ctx, int comp, double* amp, int nside, const double* b_l_out,
                                 const double* mask, int64_t npix_local, int type, double* mu) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && amp && mask && mu && npix_local >= 0, "bad arguments");
        const size_t n = (size_t)ctx->sys->ncr();
        ctx->hx.ensure(n);
        ctx->hz.ensure((size_t)std::max<int64_t>(npix_local, 1));
        CMDR_HIP_CHECK(hipMemcpy(ctx->hx.get(), amp, n * sizeof(double), hipMemcpyHostToDevice));
        CMDR_HIP_CHECK(hipMemcpy(ctx->hz.get(), mask, (size_t)npix_local * sizeof(double), hipMemcpyHostToDevice));
        if (cmdr_apply_mono_dipole_prior_dev(ctx, comp, ctx->hx.get(), nside, b_l_out, ctx->hz.get(), type, mu) != 0)
            throw cmdr::Error(g_err);
        CMDR_HIP_CHECK(hipMemcpy(amp, ctx->hx.get(), n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int cmdr_solve_dev(cmdr_ctx* ctx, const double* b, double* x, int crit, double tol, int miniter, int maxiter,
                   int check_freq, const double* x0, int* niter, double* res, int* stat) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && b && x, "bad arguments");
        CMDR_REQUIRE(crit >= 0 && crit <= 2, "crit must be 0 (residual), 1 (fixed_iter) or 2 (chisq)");
        const cmdr::SolveResult R = ctx->sys->solve(b, x, crit, tol, miniter, maxiter, check_freq, x0);
        if (niter) *niter = R.niter;
        if (stat) *stat = R.stat;
        if (res) { res[0] = R.delta_new; res[1] = R.delta0; }
    });
}
int cmdr_solve(cmdr_ctx* ctx, const double* b, double* x, int crit, double tol, int miniter, int maxiter,
               int check_freq, const double* x0, int* niter, double* res, int* stat) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && b && x, "bad arguments");
        const size_t n = (size_t)ctx->sys->ncr();
        ctx->hx.ensure(n);
        ctx->hy.ensure(n);
        ctx->hz.ensure(n);
        CMDR_HIP_CHECK(hipMemcpy(ctx->hx.get(), b, n * sizeof(double), hipMemcpyHostToDevice));
        if (x0) CMDR_HIP_CHECK(hipMemcpy(ctx->hz.get(), x0, n * sizeof(double), hipMemcpyHostToDevice));
        if (cmdr_solve_dev(ctx, ctx->hx.get(), ctx->hy.get(), crit, tol, miniter, maxiter, check_freq,
                           x0 ? ctx->hz.get() : nullptr, niter, res, stat) != 0)
            throw cmdr::Error(g_err);
        CMDR_HIP_CHECK(hipMemcpy(x, ctx->hy.get(), n * sizeof(double), hipMemcpyDeviceToHost));
    });
}

int cmdr_profile_enable(cmdr_ctx* ctx, int on) {
    return guarded([&] {
        CMDR_REQUIRE(ctx, "ctx is NULL");
        ctx->sys->set_profile(on != 0);
    });
}
int cmdr_profile_read(cmdr_ctx* ctx, double* ms_sum, long long* count) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && ms_sum && count, "bad arguments");
        ctx->sys->read_profile(ms_sum, count);
    });
}
int cmdr_profile_read_ext(cmdr_ctx* ctx, int nkinds, double* ms_sum, long long* count) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && ms_sum && count, "bad arguments");
        ctx->sys->read_profile(ms_sum, count, nkinds);
    });
}
int cmdr_problem_info(cmdr_ctx* ctx, int64_t* out) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && out, "bad arguments");
        ctx->sys->problem_info(out);
    });
}

int cmdr_problem_info_ext(cmdr_ctx* ctx, int n, int64_t* out) {
    return guarded([&] {
        CMDR_REQUIRE(ctx && out, "bad arguments");
        ctx->sys->problem_info_ext(n, out);
    });
}

int cmdr_sigma_l(const double* alm, int lmax, int nmaps, double* sigma_l) {
    return guarded([&] {
        CMDR_REQUIRE(alm && sigma_l && lmax >= 0 && nmaps >= 1 && nmaps <= 3, "bad arguments");
        CMDR_REQUIRE(cmdr_device_count() > 0, "no HIP device available: libcmdr_hip has no CPU path");
        const int64_t na = cmdr::nalm_packed(lmax);
        const int nspec = nmaps * (nmaps + 1) / 2;
        cmdr::DevBuf<double> d((size_t)na * nmaps), o((size_t)(lmax + 1) * nspec);
        CMDR_HIP_CHECK(hipMemcpy(d.get(), alm, sizeof(double) * na * nmaps, hipMemcpyHostToDevice));
        cmdr::launch_sigma_l(d.get(), na, lmax, nmaps, o.get(), nullptr);
        CMDR_HIP_CHECK(hipGetLastError());
        CMDR_HIP_CHECK(hipMemcpy(sigma_l, o.get(), sizeof(double) * (lmax + 1) * nspec, hipMemcpyDeviceToHost));
    });
}
int cmdr_sigma_l_dev(const double* alm_dev, int64_t stride, int lmax, int nmaps, double* sigma_l_dev) {
    return guarded([&] {
        CMDR_REQUIRE(alm_dev && sigma_l_dev && lmax >= 0 && nmaps >= 1 && nmaps <= 3, "bad arguments");
        cmdr::launch_sigma_l(alm_dev, stride, lmax, nmaps, sigma_l_dev, nullptr);
        CMDR_HIP_CHECK(hipGetLastError());
        CMDR_HIP_CHECK(hipStreamSynchronize(nullptr));
    });
}

int cmdr_alm_to_chain_order(const double* alm, int lmax, int nmaps, float* chain32) {
    return guarded([&] {
        CMDR_REQUIRE(alm && chain32 && lmax >= 0 && nmaps >= 1, "bad arguments");
        CMDR_REQUIRE(cmdr_device_count() > 0, "no HIP device available: libcmdr_hip has no CPU path");
        const int64_t na = cmdr::nalm_packed(lmax);
        cmdr::DevBuf<double> d((size_t)na * nmaps);
        cmdr::DevBuf<float> o((size_t)na * nmaps);
        CMDR_HIP_CHECK(hipMemcpy(d.get(), alm, sizeof(double) * na * nmaps, hipMemcpyHostToDevice));
        cmdr::launch_alm_chain(d.get(), na, o.get(), lmax, nmaps, true, nullptr);
        CMDR_HIP_CHECK(hipGetLastError());
        CMDR_HIP_CHECK(hipMemcpy(chain32, o.get(), sizeof(float) * na * nmaps, hipMemcpyDeviceToHost));
    });
}
int cmdr_alm_from_chain_order(const float* chain32, int lmax, int nmaps, double* alm) {
    return guarded([&] {
        CMDR_REQUIRE(alm && chain32 && lmax >= 0 && nmaps >= 1, "bad arguments");
        CMDR_REQUIRE(cmdr_device_count() > 0, "no HIP device available: libcmdr_hip has no CPU path");
        const int64_t na = cmdr::nalm_packed(lmax);
        cmdr::DevBuf<double> d((size_t)na * nmaps);
        cmdr::DevBuf<float> o((size_t)na * nmaps);
        CMDR_HIP_CHECK(hipMemcpy(o.get(), chain32, sizeof(float) * na * nmaps, hipMemcpyHostToDevice));
        cmdr::launch_alm_chain(d.get(), na, o.get(), lmax, nmaps, false, nullptr);
        CMDR_HIP_CHECK(hipGetLastError());
        CMDR_HIP_CHECK(hipMemcpy(alm, d.get(), sizeof(double) * na * nmaps, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
