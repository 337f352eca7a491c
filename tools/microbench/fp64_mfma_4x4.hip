// Rate of v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction, 512 flop) against
// v_mfma_f64_16x16x4_f64 (2048 flop) on gfx950.  (development aid: would a one-pair spin-2 adjoint, which fills only
// 4 of the 16 columns of the big tile, run at the full fp64 rate on the small one?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// NV independent v_fma_f64 beside NM small MFMAs per iteration: do they share the fp64 datapath like the big tile does?
template <int NV, int NM>
__global__ void __launch_bounds__(256) k_mix4(double* out, int iters, double a, double b) {
    double v[NV > 0 ? NV : 1];
    double acc[NM > 0 ? NM : 1];
#pragma unroll
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = 0.0;
    const double ma = 1e-3 * (threadIdx.x & 15), mb = 1e-3 * (threadIdx.x >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (NM > NV ? NM : NV); ++i) {
            if (i < NM) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, acc[i], 0, 0, 0);
            if (i < NV) v[i] = __builtin_fma(v[i], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// NV independent v_fmac_f64_dpp (row_newbcast operand, the synthesis kernels' accumulate) per iteration
template <int NV>
__global__ void __launch_bounds__(256) k_dpp(double* out, int iters, double a) {
    double v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 1e-3 + i;
    double g = 1e-3 * (threadIdx.x & 15), m = a;
    asm volatile("" : "+v"(g), "+v"(m));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(g), "v"(m), "n"(i & 15));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, bool SMALL>
__global__ void __launch_bounds__(256) k_mf(double* out, int iters) {
    double acc1[NM];
    d4 acc4[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) { acc1[i] = 0.0; acc4[i] = d4{0.0, 0.0, 0.0, 0.0}; }
    const double ma = 1e-3 * (threadIdx.x & 15), mb = 1e-3 * (threadIdx.x >> 4);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if (SMALL) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, acc1[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc4[i], 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc1[i] + acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
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
#define RUN(NM, SMALL)                                                                                                \
    {                                                                                                                 \
        double ms = timeit([&] { hipLaunchKernelGGL((k_mf<NM, SMALL>), dim3(nblk), dim3(256), 0, 0, out, iters); });  \
        double fm = 2.0 * NM * (SMALL ? 256.0 : 1024.0) * iters * (double)nblk * 4;                                   \
        printf("%s NM=%d: %.3f ms  %.1f TF\n", SMALL ? "4x4x4_4b " : "16x16x4  ", NM, ms, fm / ms / 1e9);             \
    }
    RUN(1, true) RUN(2, true) RUN(4, true) RUN(8, true) RUN(1, false) RUN(4, false) RUN(8, false)
#define MIX(NV, NM)                                                                                                   \
    {                                                                                                                 \
        double ms = timeit([&] { hipLaunchKernelGGL((k_mix4<NV, NM>), dim3(nblk), dim3(256), 0, 0, out, iters, 0.999, 1e-3); }); \
        double fv = 2.0 * NV * iters * (double)nblk * 256, fm = 2.0 * NM * 256.0 * iters * (double)nblk * 4;          \
        printf("mix NV=%2d NM=%d: %.3f ms   VALU %.1f TF  MFMA4 %.1f TF  sum %.1f TF\n", NV, NM, ms, fv / ms / 1e9, fm / ms / 1e9, (fv + fm) / ms / 1e9); \
    }
    {
        double ms = timeit([&] { hipLaunchKernelGGL((k_dpp<16>), dim3(nblk), dim3(256), 0, 0, out, iters, 1e-9); });
        printf("v_fmac_f64_dpp x16: %.3f ms  %.1f TF\n", ms, 2.0 * 16 * iters * (double)nblk * 256 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((k_dpp<8>), dim3(nblk), dim3(256), 0, 0, out, iters, 1e-9); });
        printf("v_fmac_f64_dpp x8:  %.3f ms  %.1f TF\n", ms, 2.0 * 8 * iters * (double)nblk * 256 / ms / 1e9);
    }
    MIX(8, 0) MIX(16, 0) MIX(0, 8) MIX(2, 8) MIX(4, 8) MIX(8, 8) MIX(16, 8)
    return 0;
}
