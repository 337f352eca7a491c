// Operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment (development aid):
//   A lane la holds 1 + la, B is one-hot at lane lb; every D lane that comes back non-zero names the A lane it met.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int CBSZ, int ABID>
__global__ void k_probe_bc(double* out) {
    const int lane = threadIdx.x;
    for (int lb = 0; lb < 64; ++lb) {
        const double a = 1.0 + lane, b = lane == lb ? 1.0 : 0.0;
        out[lb * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    }
}

__global__ void k_probe(double* out) {
    const int lane = threadIdx.x;
    for (int lb = 0; lb < 64; ++lb) {
        const double a = 1.0 + lane, b = lane == lb ? 1.0 : 0.0;
        out[lb * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    }
}

int main() {
    double* out; CK(hipMalloc(&out, sizeof(double) * 64 * 64));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out);
    static double h[64 * 64];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    // D[d] = sum over (la, lb) pairs that meet in d: print, for every B lane, the (D lane <- A lane) list
    for (int lb = 0; lb < 64; ++lb) {
        printf("B lane %2d:", lb);
        for (int d = 0; d < 64; ++d)
            if (h[lb * 64 + d] != 0.0) printf("  D%-2d<-A%-2d", d, (int)h[lb * 64 + d] - 1);
        printf("\n");
    }
#define BC(C, AB)                                                                                     \
    {                                                                                                 \
        hipLaunchKernelGGL((k_probe_bc<C, AB>), dim3(1), dim3(64), 0, 0, out);                        \
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));                                      \
        printf("cbsz=%d abid=%d\n", C, AB);                                                           \
        for (int lb = 0; lb < 64; lb += 5) {                                                          \
            printf("B lane %2d:", lb);                                                                \
            for (int d = 0; d < 64; ++d)                                                              \
                if (h[lb * 64 + d] != 0.0) printf("  D%-2d<-A%-2d", d, (int)h[lb * 64 + d] - 1);      \
            printf("\n");                                                                             \
        }                                                                                             \
    }
    BC(2, 0) BC(2, 1) BC(2, 3) BC(1, 0) BC(1, 1)
    return 0;
}
