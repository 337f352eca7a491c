// Microbenchmarks for the fp64 Legendre synthesis inner loop on gfx950 (development aid, not part of the product).
//   mode 0: pure independent v_fma_f64 chains (VALU fp64 peak / sustained clock)
//   mode 1: synth-like loop (R ring pairs, 1 map), coefficients = loop-invariant scalars (no memory)
//   mode 2: same, coefficients via scalar loads from a tiny (cache-resident) table
//   mode 3: same, coefficients via scalar loads streaming through a big per-wave table (scalar-cache misses)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NCH>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a, double b) {
    double v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[i] = __builtin_fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int R, int MODE, int UN>
__global__ void __launch_bounds__(256) k_synth(double* out, const double* __restrict__ tab, long wave_stride, int nl,
                                               double a0, double a1) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long wave = (long)blockIdx.x * 4 + wid;
    const double* __restrict__ t = tab + (MODE == 3 ? wave * wave_stride : (MODE == 4 ? (long)blockIdx.x * wave_stride : 0));
    double x[R], mc[R], mp[R], Er[R], Ei[R], Or[R], Oi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = 0.3 + 1e-3 * (threadIdx.x & 63) + 0.01 * r; mc[r] = 1e-3; mp[r] = 0; Er[r] = Ei[r] = Or[r] = Oi[r] = 0; }
    for (int l = 0; l < nl; l += 2 * UN) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            double c0r, c0i, c1r, c1i, al1, al2;
            if (MODE == 1) { c0r = a0; c0i = a1; c1r = a1; c1i = a0; al1 = 1.9; al2 = 1.95; }
            else {
                const int ll = (MODE == 2) ? ((l + 2 * u) & 63) : ((l + 2 * u) >> 1);
                c0r = t[6 * ll]; c0i = t[6 * ll + 1]; al1 = t[6 * ll + 2]; c1r = t[6 * ll + 3]; c1i = t[6 * ll + 4]; al2 = t[6 * ll + 5];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Er[r] += mc[r] * c0r; Ei[r] += mc[r] * c0i;
                double tt = al1 * x[r] * mc[r] - mp[r]; mp[r] = mc[r]; mc[r] = tt;
                Or[r] += mc[r] * c1r; Oi[r] += mc[r] * c1i;
                tt = al2 * x[r] * mc[r] - mp[r]; mp[r] = mc[r]; mc[r] = tt;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += Er[r] + Ei[r] + Or[r] + Oi[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 5: the block's 4 waves share one stream; tiles of TL l-pairs are staged into LDS by coalesced vector loads
// and consumed through broadcast ds_reads (VGPR operands).
template <int R, int TL>
__global__ void __launch_bounds__(256) k_synth_lds(double* out, const double* __restrict__ tab, long wave_stride, int nl) {
    __shared__ double tile[2][TL * 6];
    const double* __restrict__ t = tab + (long)blockIdx.x * wave_stride;
    double x[R], mc[R], mp[R], Er[R], Ei[R], Or[R], Oi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = 0.3 + 1e-3 * (threadIdx.x & 63) + 0.01 * r; mc[r] = 1e-3; mp[r] = 0; Er[r] = Ei[r] = Or[r] = Oi[r] = 0; }
    const int npair = nl / 2;
    int buf = 0;
    for (int i = threadIdx.x; i < TL * 6; i += 256) tile[0][i] = t[i];
    __syncthreads();
    for (int p0 = 0; p0 < npair; p0 += TL) {
        if (p0 + TL < npair) for (int i = threadIdx.x; i < TL * 6; i += 256) tile[buf ^ 1][i] = t[(long)(p0 + TL) * 6 + i];
        const double* tl = tile[buf];
#pragma unroll 4
        for (int p = 0; p < TL; ++p) {
            const double c0r = tl[6 * p], c0i = tl[6 * p + 1], al1 = tl[6 * p + 2], c1r = tl[6 * p + 3], c1i = tl[6 * p + 4], al2 = tl[6 * p + 5];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Er[r] += mc[r] * c0r; Ei[r] += mc[r] * c0i;
                double tt = al1 * x[r] * mc[r] - mp[r]; mp[r] = mc[r]; mc[r] = tt;
                Or[r] += mc[r] * c1r; Oi[r] += mc[r] * c1i;
                tt = al2 * x[r] * mc[r] - mp[r]; mp[r] = mc[r]; mc[r] = tt;
            }
        }
        __syncthreads();
        buf ^= 1;
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) s += Er[r] + Ei[r] + Or[r] + Oi[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> double timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const int nblk = 256 * 16;  // 16 blocks of 4 waves per CU
    double* out; CK(hipMalloc(&out, sizeof(double) * nblk * 256));
    const int nl = 2048;
    const long wave_stride = 6L * nl / 2 * 2;  // 6 doubles per l-pair... generous
    std::vector<double> h((size_t)nblk * 4 * wave_stride + 1024, 0.0);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (i % 97) + ((i % 6 == 2 || i % 6 == 5) ? 1.9 : 0.0);
    double* tab; CK(hipMalloc(&tab, h.size() * sizeof(double))); CK(hipMemcpy(tab, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        const int iters = 4096;
        double ms = timeit([&] { hipLaunchKernelGGL(k_fma<16>, dim3(nblk), dim3(256), 0, 0, out, iters, 0.999, 1e-3); });
        double fl = 2.0 * 16 * iters * (double)nblk * 256;
        printf("mode0 pure fma 16 chains: %.3f ms  %.2f TFLOP/s\n", ms, fl / ms / 1e9);
    }
#define RUN(R, MODE, UN)                                                                                           \
    {                                                                                                              \
        double ms = timeit([&] { hipLaunchKernelGGL((k_synth<R, MODE, UN>), dim3(nblk), dim3(256), 0, 0, out, tab, \
                                                    wave_stride, nl, 0.7, 0.3); });                                \
        double steps = (double)nblk * 4 * nl;  /* wave-l steps */                                                  \
        double cyc = ms * 1e-3 * 2.4e9 * 1024 / steps;                                                             \
        printf("R=%d mode=%d unroll=%dx2l: %.3f ms  -> %.1f SIMD-cycles(@2.4GHz) per wave-l  (VALU min %d), alg %.2f TFLOP/s\n", R, MODE, UN, ms, cyc, 16 * R, 8.0 * R * 64 * steps / ms / 1e9); \
    }
#define RUNL(R, TL)                                                                                                \
    {                                                                                                              \
        double ms = timeit([&] { hipLaunchKernelGGL((k_synth_lds<R, TL>), dim3(nblk), dim3(256), 0, 0, out, tab,   \
                                                    wave_stride, nl); });                                          \
        double steps = (double)nblk * 4 * nl;                                                                      \
        printf("R=%d LDS tile=%d pairs: %.3f ms -> %.1f SIMD-cycles per wave-l (VALU min %d), alg %.2f TFLOP/s\n", R, TL, ms, ms * 1e-3 * 2.4e9 * 1024 / steps, 16 * R, 8.0 * R * 64 * steps / ms / 1e9); \
    }
    RUNL(4, 64) RUNL(4, 256) RUNL(2, 64) RUNL(2, 256)
    RUN(4, 4, 1) RUN(4, 4, 2) RUN(2, 4, 1)
    RUN(4, 1, 1) RUN(4, 2, 1) RUN(4, 3, 1) RUN(4, 3, 2) RUN(4, 3, 4)
    RUN(2, 1, 1) RUN(2, 2, 1) RUN(2, 3, 1) RUN(2, 3, 2) RUN(2, 3, 4)
    RUN(1, 1, 1) RUN(1, 3, 1) RUN(1, 3, 4)
    return 0;
}
