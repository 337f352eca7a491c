// Does fp64 MFMA co-execute with fp64 VALU FMAs on gfx950, or do they share the DP datapath?  (development aid)
//   A: NV independent v_fma_f64 per iteration          B: NM independent v_mfma_f64_16x16x4_f64 per iteration
//   C: both in the same loop (independent registers).  If t(C) ~ max(t(A), t(B)) the pipes are separate and the
//   accumulate half of the Legendre adjoint could move to the matrix unit; if t(C) ~ t(A) + t(B) there is nothing to gain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, int NM>
__global__ void __launch_bounds__(256) k_mix(double* out, int iters, double a, double b) {
    double v[NV > 0 ? NV : 1];
    d4 acc[NM > 0 ? NM : 1];
#pragma unroll
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    const double ma = 1e-3 * (threadIdx.x & 15), mb = 1e-3 * (threadIdx.x >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
double timeit(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}

int main() {
    const int nblk = 256 * 8, iters = 4096;
    double* out; CK(hipMalloc(&out, sizeof(double) * nblk * 256));
#define RUN(NV, NM)                                                                                                   \
    {                                                                                                                 \
        double ms = timeit([&] { hipLaunchKernelGGL((k_mix<NV, NM>), dim3(nblk), dim3(256), 0, 0, out, iters, 0.999, 1e-3); }); \
        double fv = 2.0 * NV * iters * (double)nblk * 256, fm = 2.0 * NM * 1024.0 * iters * (double)nblk * 4;         \
        printf("NV=%2d NM=%d: %.3f ms   VALU %.1f TF  MFMA %.1f TF  sum %.1f TF\n", NV, NM, ms, fv / ms / 1e9, fm / ms / 1e9, (fv + fm) / ms / 1e9); \
    }
    RUN(16, 0) RUN(0, 1) RUN(0, 2) RUN(0, 4) RUN(0, 8) RUN(2, 4) RUN(4, 8) RUN(16, 1) RUN(16, 4)
    return 0;
}
