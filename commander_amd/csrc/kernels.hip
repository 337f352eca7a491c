// gfx950 kernels of libcmdr_hip and their launchers.  Bodies live in kernels_body.hpp.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "kernels_body.hpp"

namespace cmdr {

// ===================================================================================== Legendre stage
// grid.x = ceil(ntasks/4) (one wave task per wavefront), grid.y = nmaps.  No LDS, no barriers.
template <int R>
__global__ void __launch_bounds__(256) k_leg_synth(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                   const double* __restrict__ ast, int64_t ast_stride,
                                                   double* __restrict__ ph, int64_t ph_stride) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    leg_synth_lane<R>(A, ast + blockIdx.y * ast_stride, ph + blockIdx.y * ph_stride, m, chunk, lw, lAend,
                      threadIdx.x & 63);
}

// Adjoint: each wave reduces its 64 lanes through a private LDS tile and writes one partial column segment
// part[map][chunk][padded triangle] (complex).  Deterministic: fixed summation order, no atomics.
template <int R, bool SQUARE>
__global__ void __launch_bounds__(256) k_leg_adj(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                 const double* __restrict__ ph, int64_t ph_stride,
                                                 double* __restrict__ part, int64_t part_map_stride,
                                                 int64_t part_chunk_stride) {
    __shared__ double lds[4][16 * 65];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lmax = A.lmax;
    AdjLane<R> S;
    leg_adj_load<R, SQUARE>(A, ph + blockIdx.y * ph_stride, m, chunk, lane, S);
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    double* __restrict__ out = part + blockIdx.y * part_map_stride + chunk * part_chunk_stride + 2 * (mo - m);
    double* wl = lds[wid];
    const int col = lane & 15, qtr = lane >> 4;
    for (int l0 = lw; l0 <= lmax; l0 += kAdjL_) {
        double v[16];
        if (l0 < lAend) leg_adj_group<R, SQUARE, true>(A, al, l0, S, v);
        else            leg_adj_group<R, SQUARE, false>(A, al, l0, S, v);
#pragma unroll
        for (int j = 0; j < 16; ++j) wl[j * 65 + lane] = v[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += wl[col * 65 + qtr * 16 + i];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const int l = l0 + (col >> 1);
        if (qtr == 0 && l <= lmax) out[2 * l + (col & 1)] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int R>
static void launch_leg_synth_R(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast,
                               int64_t ast_stride, double* ph, int64_t ph_stride, int nmaps, hipStream_t s) {
    dim3 grid((ntasks + 3) / 4, nmaps);
    hipLaunchKernelGGL(k_leg_synth<R>, grid, dim3(256), 0, s, A, tasks, ntasks, ast, ast_stride, ph, ph_stride);
}
void launch_leg_synth(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, int64_t ast_stride,
                      double* ph, int64_t ph_stride, int nmaps, hipStream_t s) {
    if (ntasks == 0 || nmaps == 0) return;
    switch (A.R) {
        case 1: launch_leg_synth_R<1>(A, tasks, ntasks, ast, ast_stride, ph, ph_stride, nmaps, s); break;
        case 2: launch_leg_synth_R<2>(A, tasks, ntasks, ast, ast_stride, ph, ph_stride, nmaps, s); break;
        default: launch_leg_synth_R<4>(A, tasks, ntasks, ast, ast_stride, ph, ph_stride, nmaps, s); break;
    }
}

template <int R, bool SQ>
static void launch_leg_adj_R(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph,
                             int64_t ph_stride, double* part, int64_t pms, int64_t pcs, int nmaps, hipStream_t s) {
    dim3 grid((ntasks + 3) / 4, nmaps);
    hipLaunchKernelGGL((k_leg_adj<R, SQ>), grid, dim3(256), 0, s, A, tasks, ntasks, ph, ph_stride, part, pms, pcs);
}
void launch_leg_adj(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                    double* part, int64_t part_map_stride, int64_t part_chunk_stride, int nmaps, bool square,
                    hipStream_t s) {
    if (ntasks == 0 || nmaps == 0) return;
#define CMDR_ADJ(RR)                                                                                           \
    if (square) launch_leg_adj_R<RR, true>(A, tasks, ntasks, ph, ph_stride, part, part_map_stride,             \
                                           part_chunk_stride, nmaps, s);                                       \
    else launch_leg_adj_R<RR, false>(A, tasks, ntasks, ph, ph_stride, part, part_map_stride, part_chunk_stride, \
                                     nmaps, s);
    switch (A.R) {
        case 1: CMDR_ADJ(1) break;
        case 2: CMDR_ADJ(2) break;
        default: CMDR_ADJ(4) break;
    }
#undef CMDR_ADJ
}

// ===================================================================================== ring stage
// One workgroup = one ring pair of one map; the packed complex spectrum / pixels live in dynamic LDS.
//   MODE 0: phases -> pixels (alm2map tail)         out map = y * (mul ? mul[pix] : 1) * (weighted ? wgt : 1)
//   MODE 1: pixels -> phases (map2alm head)         in  map * (mul ? mul[pix] : 1) * (weighted ? wgt : 1)
//   MODE 2: phases -> pixels * mul[pix] -> phases   (fused Y, N^-1, Y^T of the CR matvec; map never hits HBM)
template <int MODE>
__global__ void __launch_bounds__(512) k_ring(const RingDev* __restrict__ rings, const int* __restrict__ cls,
                                              double* __restrict__ ph, int64_t ph_stride, int64_t npair_pad,
                                              double* __restrict__ map, int64_t map_stride,
                                              const double* const* __restrict__ mul, int weighted,
                                              const cd* __restrict__ tw, int log2Mmax,
                                              const cd* __restrict__ chirp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cd* buf = reinterpret_cast<cd*>(smem);
    const int pair = cls[blockIdx.x];
    const int imap = blockIdx.y;
    const RingDev d = rings[pair];
    const FftCtx c{(int)threadIdx.x, (int)blockDim.x};
    double* php = ph + imap * ph_stride;
    double* mp = map ? map + imap * map_stride : nullptr;
    const double* mu = mul ? mul[imap] : nullptr;
    const int n = d.nphi;
    const double wg = weighted ? d.wgt : 1.0;
    if (MODE == 0 || MODE == 2) {
        ring_synth_lds(buf, d, php, npair_pad, pair, tw, log2Mmax, chirp, c);
        // (every exit path of ring_synth_lds ends with a barrier)
    }
    if (MODE == 0) {
        for (int k = c.tid; k < n; k += c.nthr) {
            const cd v = buf[k];
            const double fn = wg * (mu ? mu[d.startN + k] : 1.0);
            mp[d.startN + k] = v.x * fn;
            if (d.startS >= 0) {
                const double fs = wg * (mu ? mu[d.startS + k] : 1.0);
                mp[d.startS + k] = v.y * fs;
            }
        }
        return;
    }
    if (MODE == 1) {
        for (int k = c.tid; k < n; k += c.nthr) {
            const double fn = wg * (mu ? mu[d.startN + k] : 1.0);
            cd v = {mp[d.startN + k] * fn, 0.0};
            if (d.startS >= 0) {
                const double fs = wg * (mu ? mu[d.startS + k] : 1.0);
                v.y = mp[d.startS + k] * fs;
            }
            buf[k] = v;
        }
        __syncthreads();
    }
    if (MODE == 2) {
        for (int k = c.tid; k < n; k += c.nthr) {
            cd v = buf[k];
            v.x *= mu[d.startN + k];
            v.y = d.startS >= 0 ? v.y * mu[d.startS + k] : 0.0;
            buf[k] = v;
        }
        __syncthreads();
    }
    ring_anal_lds(buf, d, tw, log2Mmax, chirp, c);
    ring_store_phases(buf, d, php, npair_pad, pair, c);
}

void launch_ring(int mode, const RingDev* rings, const int* cls, int ncls, int log2M, double* ph,
                 int64_t ph_stride, int64_t npair_pad, double* map, int64_t map_stride, const double* const* mul,
                 int weighted, const cd* tw, int log2Mmax, const cd* chirp, int nmaps, hipStream_t s) {
    if (ncls == 0 || nmaps == 0) return;
    const size_t lds = (size_t)sizeof(cd) << log2M;
    const int nthr = log2M >= 13 ? 512 : 256;
    dim3 grid(ncls, nmaps);
#define CMDR_RING(MM)                                                                                            \
    do {                                                                                                         \
        static bool attr_set = false;                                                                            \
        if (!attr_set) {                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<MM>),                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
            attr_set = true;                                                                                     \
        }                                                                                                        \
        hipLaunchKernelGGL(k_ring<MM>, grid, dim3(nthr), lds, s, rings, cls, ph, ph_stride, npair_pad, map,      \
                           map_stride, mul, weighted, tw, log2Mmax, chirp);                                      \
    } while (0)
    if (mode == 0) CMDR_RING(0);
    else if (mode == 1) CMDR_RING(1);
    else CMDR_RING(2);
#undef CMDR_RING
}

// ===================================================================================== a_lm streaming kernels
// All run on an (l, m) grid: blockIdx.y = m, l = m + blockIdx.x*256 + threadIdx.x.

// Commander real-packed a_lm (one column) -> padded-triangle complex stream, times cnorm * kappa_m * scale.
// kappa_m = 1/sqrt2 for m>0 (Hermitian pair construction in the ring stage), 1 for m=0.
__global__ void k_alm_to_stream(const double* __restrict__ alm, int64_t alm_stride, double* __restrict__ ast,
                                int64_t ast_stride, const double* __restrict__ cnorm, int lmax) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax + 1) return;
    alm_to_stream_elem(alm + blockIdx.z * alm_stride, ast + blockIdx.z * ast_stride, cnorm, lmax, m, l);
}
void launch_alm_to_stream(const double* alm, int64_t alm_stride, double* ast, int64_t ast_stride,
                          const double* cnorm, int lmax, int nmaps, hipStream_t s) {
    dim3 grid((lmax + 2 + 255) / 256, lmax + 1, nmaps);
    hipLaunchKernelGGL(k_alm_to_stream, grid, dim3(256), 0, s, alm, alm_stride, ast, ast_stride, cnorm, lmax);
}

// partial columns -> Commander real-packed a_lm: alm = kappa'_m * cnorm * sum_chunks part ; kappa' = sqrt2 (m>0).
__global__ void k_part_to_alm(const double* __restrict__ part, int64_t part_map_stride, int64_t part_chunk_stride,
                              int nchunk, double* __restrict__ alm, int64_t alm_stride,
                              const double* __restrict__ cnorm, int lmax) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax) return;
    part_to_alm_elem(part + blockIdx.z * part_map_stride, part_chunk_stride, nchunk, alm + blockIdx.z * alm_stride,
                     cnorm, lmax, m, l);
}
void launch_part_to_alm(const double* part, int64_t pms, int64_t pcs, int nchunk, double* alm, int64_t alm_stride,
                        const double* cnorm, int lmax, int nmaps, hipStream_t s) {
    dim3 grid((lmax + 1 + 255) / 256, lmax + 1, nmaps);
    hipLaunchKernelGGL(k_part_to_alm, grid, dim3(256), 0, s, part, pms, pcs, nchunk, alm, alm_stride, cnorm, lmax);
}

}  // namespace cmdr
