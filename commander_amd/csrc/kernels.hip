// gfx950 kernels of libcmdr_hip and their launchers.  Bodies live in kernels_body.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"
#include "kernels_body.hpp"

namespace cmdr {

// ===================================================================================== Legendre stage
// Synthesis: one wave task (one m x 64*R ring pairs) per wavefront, 4 per workgroup, NB maps per wave sharing the
// recursion.  No LDS, no barriers: the wave-uniform operands (alpha_l and the a_lm stream) come in through the
// scalar unit.  Measured on MI355X (tools/microbench/fp64_synth.hip): fp64 FMA sustains ~61 TFLOP/s (clock ~1.9 GHz
// under load), scalar-cache hits are free, scalar-cache MISSES sustain only ~1.5 B/clk/CU, and LDS broadcast reads
// ~4 B/clk/CU -- so the stream bytes per VALU cycle, not the VALU itself, bound how many maps can share one
// recursion.  (R = 4, NB = 3) is the measured optimum; an LDS-staged variant was slower (register pressure).
template <int R, int NB>
__global__ void __launch_bounds__(256) k_leg_synth(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                   const double* __restrict__ ast, int nbs, int k0, int rep,
                                                   double* __restrict__ ph, int64_t ph_stride) {
    // rep batches of NB maps in one launch, task-major so that the longest tasks of every batch start first
    // (small shards: a launch per batch would each wait for its own longest wave)
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = (blockIdx.x / rep) * 4 + wid;
    k0 += (blockIdx.x % rep) * NB;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    leg_synth_lane<R, NB>(A, ast, nbs, k0, ph, ph_stride, m, chunk, lw, lAend, threadIdx.x & 63);
}

// Synthesis, workgroup form: the 4 waves of a workgroup are 4 adjacent chunks of ONE m (LegArgs::wg).  The column's
// coefficients (NB complex a_lm entries and alpha per l) are fetched once per workgroup by coalesced vector loads,
// staged through a double-buffered LDS tile of kTileL l values and read back as LDS broadcasts (uniform VGPR
// operands): the scalar-cache miss bandwidth (~1.5 B/clk/CU, measured) that bounds k_leg_synth no longer enters.
constexpr int kTileL = 32;
// acc += a * g(lane Q of this lane's 16-lane row): gfx90a+ let fp64 VALU ops take src0 through the DPP row broadcast,
// so the operand that is shared by the 16 l-rows of one pair costs no instruction and no LDS read of its own.
// (Hazard note: a VGPR written by a VALU instruction must not be read through DPP by one of the next two instructions;
// the hazard recogniser cannot see inside inline asm.  Both users pass registers filled by LDS / global loads long
// before -- the parity tests at full size are the check that a future compiler does not slip a copy in between.)
template <int Q>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double g, double a) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(a), "n"(Q));
}

// One l of the synthesis for R ring pairs per lane and NB maps, coefficients through DPP row broadcasts: C[c] holds, in
// lane j of every 16-lane row, column c of tile row lb16 + j (a~ re / im of map c / 2), so step J takes lane J of the
// row inside the FMA; alpha_{l+1} comes through the scalar unit.  Same operations in the same order as the LDS-broadcast loop.
// a4 / a4n: alpha of the first four steps, fetched by the previous block (its scalar load would otherwise be waited for at
// the top of every block), and the same for the next block, requested at step 1 together with this block's own.
template <int R, int NB, bool INJECT, int J>
__device__ __forceinline__ void synth_steps16(const double (&C)[2 * NB], const double* __restrict__ alp /* alpha_{lb16+1+j} */,
                                              int lb16, const double (&x)[R], double (&mc)[R],
                                              double (&mp)[R], const double (&sc)[R], const double (&sp)[R],
                                              const int (&ls)[R], double (&Er)[R][NB], double (&Ei)[R][NB],
                                              double (&Or)[R][NB], double (&Oi)[R][NB], const double (&a4)[4],
                                              double (&a4n)[4]) {
    if constexpr (J < 16) {
        if constexpr (J == 1) {
            __builtin_amdgcn_sched_barrier(0);             // (... or hoist it above the waits for the tile rows)
#pragma unroll
            for (int i = 0; i < 4; ++i) a4n[i] = alp[16 + i];
            __builtin_amdgcn_sched_barrier(0);             // (the scheduler would sink the request to the block's end)
        }
        double alj;
        if constexpr (J < 4) alj = a4[J]; else alj = alp[J];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (INJECT) if (ls[r] == lb16 + J) { mc[r] = sc[r]; mp[r] = sp[r]; }   // phase A: lanes still switch on
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if constexpr ((J & 1) == 0) {
                    fmac_row_bcast<J>(Er[r][k], C[2 * k], mc[r]);
                    fmac_row_bcast<J>(Ei[r][k], C[2 * k + 1], mc[r]);
                } else {
                    fmac_row_bcast<J>(Or[r][k], C[2 * k], mc[r]);
                    fmac_row_bcast<J>(Oi[r][k], C[2 * k + 1], mc[r]);
                }
            }
            const double ax = alj * x[r];                      // alpha_{l+1} through the scalar unit
            const double tt = ax * mc[r] - mp[r];
            mp[r] = mc[r];
            mc[r] = tt;
        }
        synth_steps16<R, NB, INJECT, J + 1>(C, alp, lb16, x, mc, mp, sc, sp, ls, Er, Ei, Or, Oi, a4, a4n);
    }
}

// PREP: the coefficient stream is not read from memory but formed while the tile is staged (k_band_prep folded in)
// DPPC: the tile is consumed 16 l at a time through DPP row broadcasts (synth_steps16) instead of one LDS broadcast read
// per coefficient and l
// UNI (with DPPC): the plan starts every 64-pair lane block at one l == m (mod 32), so seeds are injected once per block
// at a 16-l block boundary and the stepping code carries no per-l start test at all
template <int R, int NB, bool PREP, bool DPPC, bool UNI>
__global__ void __launch_bounds__(256, 3) k_leg_synth_wg(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                      const double* __restrict__ ast, int nbs, int k0, int rep,
                                                      double* __restrict__ ph, int64_t ph_stride, PrepDev P) {
    // doubles per l: NB x (re, im) [, alpha_{l+1}, pad: LDS form only].  DPP form: lane j of a 16-lane row reads tile row j
    // with ds_read_b128, so the row stride decides the banking: 16 B, 48 B and 80 B strides are conflict-free, 32 B (NB = 2)
    // is 2-way and 64 B (NB = 4) 4-way (7.4e7 conflict cycles per launch, profiles/r02_pmc_sq_summary.txt).  NB = 2 takes a
    // pad slot; NB = 4 does NOT: 80-byte rows make the tile 320 doubles, i.e. two staging loads per thread instead of one,
    // and the launch 1.37 instead of 1.23 ms (measured, round 3) -- the conflicts are cheaper than the second load
    constexpr int ROW = DPPC ? (NB == 2 ? 2 * NB + 2 : 2 * NB) : 2 * NB + 2;
    constexpr int NE = kTileL * ROW;                    // doubles per tile
    constexpr int NLD = (NE + 255) / 256;               // global loads per thread and tile
    __shared__ __attribute__((aligned(16))) double tile[2][NE];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int tb = blockIdx.x / rep;
    k0 += (blockIdx.x % rep) * NB;
    const WaveTask T = tasks[tb * 4 + wid];
    const int m = __builtin_amdgcn_readfirstlane(tasks[tb * 4].m);
    int lw0 = tasks[tb * 4].lw;
#pragma unroll
    for (int i = 1; i < 4; ++i) lw0 = min(lw0, tasks[tb * 4 + i].lw);
    lw0 = __builtin_amdgcn_readfirstlane(lw0);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lmax = A.lmax;
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ as = ast + 2 * ((int64_t)nbs * (mo - m) + k0);
    const int64_t ls2 = 2 * (int64_t)nbs;
    // per-lane state (as leg_synth_lane)
    double x[R], mc[R], mp[R], sc[R], sp[R];
    double Er[R][NB], Ei[R][NB], Or[R][NB], Oi[R][NB];
    int ls[R];
    const int base = (chunk < 0 ? 0 : chunk) * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x[r] = A.x[p];
        ls[r] = chunk < 0 ? 0x7fffffff : A.ls[idx];
        sc[r] = A.seedc[idx];
        sp[r] = A.seedp[idx];
        mc[r] = mp[r] = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) Er[r][k] = Ei[r][k] = Or[r][k] = Oi[r][k] = 0.0;
    }
    int lwr[R];                                          // UNI: the common start of lane block r (wave-uniform)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int v = ls[r];
        if (UNI) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        }
        lwr[r] = __builtin_amdgcn_readfirstlane(v);
    }
    // UNI: the seeds as the block's start will take them (zero in the lanes that never start), formed HERE: their first
    // use inside the tile loop would leave the wait for these loads there, as a vmcnt(0) that also waits, tile after
    // tile, for the tile loads issued just before it
    double scv[R], spv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool on = UNI && ls[r] == lwr[r];
        scv[r] = on ? sc[r] : 0.0;
        spv[r] = on ? sp[r] : 0.0;
    }
    // PREP: per (map of this batch, component) where the component's (l = 0, m) entry sits in sx (the packed index is
    // linear in l within one m), where its weight row starts, and up to which l it contributes
    struct PrepTerm { long long sxo, wo; int lmaxc, pad; };
    __shared__ PrepTerm terms[PREP ? 5 * 8 : 1];
    const int sl = m == 0 ? 1 : 2;
    const int64_t na = (int64_t)(lmax + 1) * (lmax + 1);
    const int64_t gbase = d_packed_index(lmax, 0, m);
    if (PREP) {
        if ((int)threadIdx.x < NB * P.ncomp) {
            const int k = threadIdx.x / P.ncomp, c = threadIdx.x - k * P.ncomp, bm = k0 + k;
            const CompDev C = P.comps[c];
            const int st = P.bm_stokes[bm];
            PrepTerm T;
            T.lmaxc = st < C.nmaps ? C.lmax : -1;
            T.sxo = C.pos + (int64_t)st * C.nalm + d_packed_index(C.lmax, 0, m);
            T.wo = ((int64_t)bm * P.ncomp + c) * (lmax + 1);
            T.pad = 0;
            terms[threadIdx.x] = T;
        }
        __syncthreads();
    }
    // One diffuse component and no varying-mixing term (the headline case): everything about a thread's tile elements
    // but l is fixed, so the three addresses are formed once
    const bool prep1 = PREP && P.ncomp == 1 && !P.extra;
    const double* pw[NLD];
    const double* psx[NLD];
    int plm[NLD], prow[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        const int row = e / ROW, col = e - row * ROW;
        prow[i] = row;
        pw[i] = psx[i] = nullptr;
        plm[i] = -1;
        if (PREP && prep1 && e < NE && col < 2 * NB) {
            const PrepTerm T = terms[col >> 1];
            pw[i] = P.w + T.wo;
            psx[i] = P.sx + T.sxo + (col & 1);
            plm[i] = ((col & 1) && m == 0) ? -1 : min(T.lmaxc, lmax);
        }
    }
    const double* __restrict__ pcn = PREP ? P.cnorm + (mo - m) : nullptr;
    const double kap = m == 0 ? 1.0 : 0.70710678118654752440;
    // tile element e -> (row = l - lb, col): col < 2 NB: stream double, col == 2 NB: alpha_{l+1}
    // prep1: the loads of a tile are issued one tile ahead and their arithmetic is done when the tile is stored
    // (fetch_finish, at the top of the loop): done right behind the loads it puts a vmcnt(0) wait -- a trip to L2 / HBM --
    // in front of the first 16 l of every tile
    double raw[NLD][3];
    auto fetch = [&](int lb, double* v) {
        if (PREP && prep1) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int l = lb + prow[i];
                raw[i][0] = raw[i][1] = raw[i][2] = 0.0;
                if (l <= plm[i]) {
                    raw[i][0] = pw[i][l];
                    raw[i][1] = psx[i][sl * l];
                    raw[i][2] = pcn[l];
                } else if (!DPPC && (int)threadIdx.x + 256 * i < NE && (int)threadIdx.x + 256 * i - prow[i] * ROW == 2 * NB &&
                           l <= lmax + 1) {
                    raw[i][0] = al[l + 1];               // alpha column of the LDS form: passes through fetch_finish
                    raw[i][1] = 1.0;
                    raw[i][2] = 1.0 / kap;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = (int)threadIdx.x + 256 * i;
            const int row = e / ROW, col = e - row * ROW;
            const int l = lb + row;
            double val = 0.0;
            if (e < NE && l <= lmax + 1) {
                if (col < 2 * NB) {
                    if (PREP) {   // band_prep_part with the per-(map, component) constants from the LDS table
                        const int k = col >> 1, part = col & 1;
                        if (l <= lmax && !(part && m == 0)) {
                            double v = 0.0;
                            if (P.extra) v = P.extra[(int64_t)(k0 + k) * na + gbase + sl * l + part];
                            for (int c = 0; c < P.ncomp; ++c) {
                                const PrepTerm T = terms[k * P.ncomp + c];
                                if (l > T.lmaxc) continue;
                                const double wc = P.w[T.wo + l];
                                const double t = P.sx[T.sxo + sl * l + part];
                                if (wc != 0.0) v += wc * t;
                            }
                            val = v * (P.cnorm[mo - m + l] * kap);
                        }
                    } else {
                        val = as[ls2 * l + col];
                    }
                } else if (!DPPC && col == 2 * NB) {
                    val = al[l + 1];
                }
            }
            v[i] = val;
        }
    };
    auto fetch_finish = [&](double* v) {
        if (PREP && prep1) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                double a = 0.0;
                if (raw[i][0] != 0.0) a += raw[i][0] * raw[i][1];
                v[i] = a * (raw[i][2] * kap);
            }
        }
    };
    double pre[NLD];
    const int ntile = (lmax - lw0) / kTileL + 1;
    // UNI: alpha of the first four steps of the next 16-l block (synth_steps16).  The blocks a wave runs are consecutive
    // from the first one with lb16 + 16 > lw, so the first set is fetched here and every later one by the block before
    double a4[4];
    {
        const int first = lw0 + ((max(lw, lw0) - lw0) & ~15);
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = al[min(first, lmax) + 1 + i];
    }
    fetch(lw0, pre);
    for (int t = 0; t < ntile; ++t) {
        double* cur = tile[t & 1];
        __builtin_amdgcn_sched_barrier(0);
        fetch_finish(pre);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = (int)threadIdx.x + 256 * i;
            if (e < NE) cur[e] = pre[i];
        }
        __syncthreads();                                  // tile t visible; tile t-1 (same buffer as t+1) fully consumed
        const int lb = lw0 + t * kTileL;
        if (t + 1 < ntile) fetch(lb + kTileL, pre);       // in flight while this tile is consumed
        if (chunk < 0 || lb + kTileL <= lw) continue;     // this wave has not started yet (wave-uniform)
        if (DPPC) {
            // 16-l blocks; a block that starts below lw runs with mu = 0 until the seeds are injected, rows beyond
            // lmax + 1 hold zero coefficients (fetch), so neither end needs a bound
            if (UNI) {
#pragma unroll 1
                for (int hb = 0; hb < kTileL; hb += 16) {
                    const int lb16 = lb + hb;
                    if (lb16 + 16 <= lw || lb16 > lmax) continue;
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (lb16 == lwr[r]) { mc[r] = scv[r]; mp[r] = spv[r]; }   // wave-uniform: this block's lanes switch on here
                    double C[2 * NB];
                    const double* __restrict__ crow = cur + (hb + (lane & 15)) * ROW;
#pragma unroll
                    for (int c = 0; c < 2 * NB; ++c) C[c] = crow[c];
                    double a4n[4];
                    synth_steps16<R, NB, false, 0>(C, al + lb16 + 1, lb16, x, mc, mp, sc, sp, ls, Er, Ei, Or, Oi, a4, a4n);
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = a4n[i];
                }
                continue;
            }
            // (two loops one after the other, like phases A and B below: both bodies inside one loop would make the
            // register allocator copy the 4 R NB accumulators between them on every pass)
            int hb = 0;
#pragma unroll 1
            for (; hb < kTileL && lb + hb < lAend; hb += 16) {
                const int lb16 = lb + hb;
                if (lb16 + 16 <= lw || lb16 > lmax) continue;
                double C[2 * NB];
                const double* __restrict__ crow = cur + (hb + (lane & 15)) * ROW;
#pragma unroll
                for (int c = 0; c < 2 * NB; ++c) C[c] = crow[c];
                {
                    double b4[4], b4n[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) b4[i] = al[lb16 + 1 + i];
                    synth_steps16<R, NB, true, 0>(C, al + lb16 + 1, lb16, x, mc, mp, sc, sp, ls, Er, Ei, Or, Oi, b4, b4n);
                }
            }
#pragma unroll 1
            for (; hb < kTileL; hb += 16) {
                const int lb16 = lb + hb;
                if (lb16 + 16 <= lw || lb16 > lmax) continue;
                double C[2 * NB];
                const double* __restrict__ crow = cur + (hb + (lane & 15)) * ROW;
#pragma unroll
                for (int c = 0; c < 2 * NB; ++c) C[c] = crow[c];
                {
                    double b4[4], b4n[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) b4[i] = al[lb16 + 1 + i];
                    synth_steps16<R, NB, false, 0>(C, al + lb16 + 1, lb16, x, mc, mp, sc, sp, ls, Er, Ei, Or, Oi, b4, b4n);
                }
            }
            continue;
        }
        const int lend = min(lb + kTileL, lmax + 1);
        int l = max(lb, lw);
        for (; l < lend && l < lAend; l += 2) {           // Phase A: lanes switch on at their own ls
            const double* __restrict__ c0 = cur + (l - lb) * ROW;
            const double* __restrict__ c1 = c0 + ROW;
            const double al1 = c0[2 * NB], al2 = c1[2 * NB];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (ls[r] == l) { mc[r] = sc[r]; mp[r] = sp[r]; }
#pragma unroll
                for (int k = 0; k < NB; ++k) { Er[r][k] += mc[r] * c0[2 * k]; Ei[r][k] += mc[r] * c0[2 * k + 1]; }
                double tt = al1 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = tt;
                if (ls[r] == l + 1) { mc[r] = sc[r]; mp[r] = sp[r]; }
#pragma unroll
                for (int k = 0; k < NB; ++k) { Or[r][k] += mc[r] * c1[2 * k]; Oi[r][k] += mc[r] * c1[2 * k + 1]; }
                tt = al2 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = tt;
            }
        }
        for (; l < lend; l += 2) {                        // Phase B: pure recursion + accumulate
            const double* __restrict__ c0 = cur + (l - lb) * ROW;
            const double* __restrict__ c1 = c0 + ROW;
            const double al1 = c0[2 * NB], al2 = c1[2 * NB];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int k = 0; k < NB; ++k) { Er[r][k] += mc[r] * c0[2 * k]; Ei[r][k] += mc[r] * c0[2 * k + 1]; }
                double tt = al1 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = tt;
#pragma unroll
                for (int k = 0; k < NB; ++k) { Or[r][k] += mc[r] * c1[2 * k]; Oi[r][k] += mc[r] * c1[2 * k + 1]; }
                tt = al2 * x[r] * mc[r] - mp[r];
                mp[r] = mc[r];
                mc[r] = tt;
            }
        }
    }
    if (chunk < 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int p = base + r * 64;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            double* o = ph + (k0 + k) * ph_stride + d_phidx(A.lmax + 1, p, m);
            o[0] = Er[r][k] + Or[r][k];
            o[1] = Ei[r][k] + Oi[r][k];
            o[2] = Er[r][k] - Or[r][k];
            o[3] = Ei[r][k] - Oi[r][k];
        }
    }
}

// ---- wave-wide reduction of 16 values per lane -------------------------------------------------------------
// v[0..15] per lane -> on return the lanes with (lane & 3) == 0 hold sum_{64 lanes} v[lane >> 2].
// Two register butterfly steps with the gfx950 row / half swaps (v_permlane32_swap, v_permlane16_swap) fold the four
// 16-lane rows into one, leaving 4 values per lane; only those go through LDS (4 ds_write_b64 + 2 ds_read_b128 per
// lane instead of 16 + 16: the LDS pipe, not the VALU, bounded the first version of this kernel); the last factor
// of 4 is two quad-permute DPP steps.  Fixed summation order -> deterministic.
constexpr int kRedPitch = 66;                 // doubles; 33 x 16 B: conflict-free ds_read_b128 for the read pattern below
constexpr int kRedTile = 4 * kRedPitch;       // doubles per tile; two tiles per wave (double buffer)

__device__ inline void swap_halves(double& a, double& b) {   // a <- {a.lo32lanes, b.lo32lanes}, b <- {a.hi, b.hi}
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto r = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    alo = r[0]; blo = r[1];
    r = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    ahi = r[0]; bhi = r[1];
    a = __hiloint2double(ahi, alo);
    b = __hiloint2double(bhi, blo);
}
__device__ inline void swap_rows(double& a, double& b) {     // odd rows of a <-> even rows of b (row = 16 lanes)
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto r = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    alo = r[0]; blo = r[1];
    r = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    ahi = r[0]; bhi = r[1];
    a = __hiloint2double(ahi, alo);
    b = __hiloint2double(bhi, blo);
}
template <int CTRL>
__device__ inline double quad_perm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ inline double wave_reduce16(const double* v, double* tile, int lane) {
    double u[8], t[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {          // lanes < 32: value 2k, lanes >= 32: value 2k+1 (each summed over i, i^32)
        double a = v[2 * k], b = v[2 * k + 1];
        swap_halves(a, b);
        u[k] = a + b;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // row r of t[q]: value 4q + {0,2,1,3}[r], summed over the 4 rows
        double a = u[2 * q], b = u[2 * q + 1];
        swap_rows(a, b);
        t[q] = a + b;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[q * kRedPitch + lane] = t[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane = 16 q + 4 rr + part reads columns 4 part .. 4 part + 3 of value 4q + rr, which lives in row perm(rr)
    const int q = lane >> 4, rr = (lane >> 2) & 3, part = lane & 3;
    const int row = ((rr & 1) << 1) | (rr >> 1);
    const double* p = tile + q * kRedPitch + 16 * row + 4 * part;
    double s = (p[0] + p[1]) + (p[2] + p[3]);
    s += quad_perm<0xB1>(s);               // lane ^ 1
    s += quad_perm<0x4E>(s);               // lane ^ 2
    return s;
}

// ---- alpha / beta of the next l group through the scalar unit, requested by hand ------------------------------------
// The VALU adjoint kernels store a partial column entry per l group.  Behind a global store inside the loop the compiler
// no longer proves the recursion tables unclobbered, fetches alpha_{l+1} (beta_{l+1}) with VECTOR loads and waits for
// them a few instructions later: one trip to L2 per group of 4 / 8 l, in front of its arithmetic.  These helpers issue the
// s_load themselves, one group ahead; the values are taken out (sload_wait) at the bottom of the group the load flew
// behind, so what is carried around the loop is always data that has arrived.
typedef int sgpr8 __attribute__((ext_vector_type(8)));
typedef int sgpr16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ sgpr8 sload_d4(const double* p) {      // p wave-uniform
    sgpr8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(r) : "s"(p));
    return r;
}
__device__ __forceinline__ sgpr16 sload_d8(const double* p) {
    sgpr16 r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(r) : "s"(p));
    return r;
}
// `after`: a value of the group just finished, so that the wait cannot be scheduled in front of it
__device__ __forceinline__ void sload_wait(sgpr8& a, sgpr8& b, double& after) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+v"(after));
}
__device__ __forceinline__ void sload_wait(sgpr16& a, double& after) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+v"(after));
}
template <class V>
__device__ __forceinline__ double sgpr_double(const V& r, int j) { return __hiloint2double(r[2 * j + 1], r[2 * j]); }

// Adjoint: each wave reduces its 64 lanes (wave_reduce16) and writes one partial column segment
// part[map][chunk][padded triangle] (complex).  Deterministic: fixed summation order, no atomics.
template <int R, int NB, bool SQUARE>
__global__ void __launch_bounds__(256) k_leg_adj(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                 const double* __restrict__ ph, int64_t ph_stride, int k0, int rep,
                                                 double* __restrict__ part, int64_t part_map_stride,
                                                 int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double lds[4][2 * kRedTile];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int t = (blockIdx.x / rep) * 4 + wid;      // rep batches of NB maps per launch, task-major (see k_leg_synth)
    k0 += (blockIdx.x % rep) * NB;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lmax = A.lmax;
    AdjLane<R, NB> S;
    leg_adj_load<R, NB, SQUARE>(A, ph, ph_stride, k0, m, chunk, lane, S);
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    double* __restrict__ out0 = part + chunk * part_chunk_stride + 2 * (mo - m);
    double* wl = lds[wid];
    const int id = lane >> 2;              // value this lane ends up with: l = l0 + id/2, (re, im) = id & 1
    int buf = 0;
    double a8[kAdjL_];                     // alpha_{l0 + 1 + j} of the current group (sload_d8: see above)
    {
        sgpr16 a = sload_d8(al + lw + 1);
        double dummy = 0.0;
        sload_wait(a, dummy);
#pragma unroll
        for (int j = 0; j < kAdjL_; ++j) a8[j] = sgpr_double(a, j);
    }
    for (int l0 = lw; l0 <= lmax; l0 += kAdjL_) {
        sgpr16 an = sload_d8(al + l0 + kAdjL_ + 1);       // (table slack behind lmax: plan_tables.hpp ntrip)
        double w[kAdjL_][R];
        if (l0 < lAend) leg_adj_mu_group_v<R, NB, SQUARE, true>(a8, l0, S, w);
        else            leg_adj_mu_group_v<R, NB, SQUARE, false>(a8, l0, S, w);
        const int l = l0 + (id >> 1);
        double last = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            double v[16];
            leg_adj_products<R, NB>(S, w, k, v);
            const double s = wave_reduce16(v, wl + buf * kRedTile, lane);   // double-buffered: one barrier per use
            buf ^= 1;
            if ((lane & 3) == 0 && l <= lmax) out0[(k0 + k) * part_map_stride + 2 * l + (id & 1)] = s;
            last = s;
        }
        sload_wait(an, last);
#pragma unroll
        for (int j = 0; j < kAdjL_; ++j) a8[j] = sgpr_double(an, j);
    }
}

// ---- Adjoint on the matrix unit ------------------------------------------------------------------------------------
// a_lm = sum_pairs mu_l(x_pair) G_m(pair) IS a dense contraction over ring pairs once mu is laid out with l across
// lanes: D[l][col] += A[l][pair] B[pair][col] with col = (map, re/im) -- v_mfma_f64_16x16x4_f64 does the multiply-
// accumulate AND the sum over lanes that costs the VALU kernel above ~40 % of its instructions (wave_reduce16).
//   * one workgroup = 2 waves = one wave task (m, 256 ring pairs); wave w owns 128 of the pairs as two 64-pair blocks.
//   * recursion on the VALU, lane = ring pair, 32 consecutive l per group; mu goes through a wave-private LDS tile
//     [32 l][64 pairs] (pitch 65 doubles: conflict-free for the lane = pair writes and for the transposed reads) and
//     comes back as the A operand: lane i holds A[row = i & 15][k = i >> 4] = mu_{l0 + 2 row + parity}(pair 4q + k).
//   * B operands stay in registers for the whole task: lane i holds G[pair 4q + (i >> 4)][col = i & 15], once as
//     N + S (rows with even l - m) and once as N - S (odd): the north/south symmetry halves the contraction length.
//   * D (rows = 16 l of one parity, cols = 8 maps x (re, im)) accumulates over the wave's 32 pair quads; the two waves
//     exchange halves through LDS (wave 0 finishes the even rows, wave 1 the odd ones) and write the same
//     part[map][chunk][padded triangle] layout as k_leg_adj.  Fixed summation order -> deterministic.
// fp64 MFMA and fp64 VALU share one datapath on gfx950 (tools/microbench/fp64_mfma_coexec.hip), so this is not more
// flops per clock -- it removes the cross-lane reduction and lets 8 maps share one recursion (2/8 + 2 instead of
// 2/3 + 2 + 1.7-2.1 VALU-equivalents per (pair, l, map)).
typedef double mx_d4 __attribute__((ext_vector_type(4)));
constexpr int kMxL = 32;       // l per group (16 rows of each parity)
constexpr int kMxPitch = 65;   // doubles per tile row

// 32 recursion steps of one 64-pair block; mu_l of every step goes to the tile row of l.  No bound on l: the alpha table
// has slack behind its last column (plan_tables.hpp ntrip), rows beyond lmax hold finite-or-not garbage that only reaches
// D rows which are never stored (MFMA rows are independent).
template <bool INJECT>
__device__ __forceinline__ void mx_recur(const double* __restrict__ al, int l0, double x, double& mc, double& mp,
                                         double sc, double sp, int ls, double* __restrict__ trow) {
#pragma unroll
    for (int j = 0; j < kMxL; ++j) {
        const double al1 = al[l0 + j + 1];
        if (INJECT) if (ls == l0 + j) { mc = sc; mp = sp; }
        trow[j * kMxPitch] = mc;
        const double t = al1 * x * mc - mp;
        mp = mc;
        mc = t;
    }
}

// NR sub-blocks of 64 ring pairs per wave, 2 waves per task: NR = 2 -> tasks of 256 pairs (plans with R = 4),
// NR = 1 -> 128 pairs (R = 2: ring-sharded ranks with few pairs, where one task per m would leave the longest column,
// m = 0, as the critical path of the launch).  The sub-blocks of a 256-pair task are dealt 0,3 | 1,2 to the two waves
// (polar + equatorial against the two middle ones) and each wave skips the 32-l groups that lie below every start of
// a sub-block, so the (m, ring) cut is honoured per 64 pairs, not per task.
// X9: a ninth map (slot k0 + 8) rides along on the VALU: while the A operands of a 64-pair block are in registers
// (lane = (l row, pair kq of quad q)), four FMAs per MFMA step multiply them with that map's N+S / N-S phases.  Those
// sit in 8 registers per block, lane (row q', kq) holding pair 4 q' + kq, so that step q needs lane q of every 16-lane
// row: the DPP row broadcast of v_fmac_f64 delivers it inside the FMA.  The sum over the four pair lanes of a row
// closes with two cross-lane steps per 32-l group.  One map more for 1/8 more work on the shared fp64 datapath instead
// of a VALU launch of its own with its own recursion and wave-wide reductions.
template <int NR, bool X9>
__global__ void __launch_bounds__(128) k_leg_adj_mx(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                    const double* __restrict__ ph, int64_t ph_stride, int k0, int nb,
                                                    double* __restrict__ part, int64_t part_map_stride,
                                                    int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double tile[2][kMxL * kMxPitch];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= ntasks) return;
    const WaveTask T = tasks[blockIdx.x];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lmax = A.lmax;
    int pb[NR];                  // first pair of this wave's sub-blocks
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sub = NR == 1 ? wid : (wid == 0 ? 3 * r : 1 + r);
        pb[r] = chunk * (128 * NR) + sub * 64;
    }
    double x[NR], mc[NR], mp[NR], sc[NR], sp[NR];
    int ls[NR], lwr[NR], lhi[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = pb[r] + lane;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x[r] = A.x[p];
        ls[r] = A.ls[idx];
        sc[r] = A.seedc[idx];
        sp[r] = A.seedp[idx];
        mc[r] = mp[r] = 0.0;
        int v = ls[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        lwr[r] = __builtin_amdgcn_readfirstlane(v);
        v = ls[r] == 0x3fffffff ? -1 : ls[r];               // last start among the lanes that start at all
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        lhi[r] = __builtin_amdgcn_readfirstlane(v);
    }
    const int kq = lane >> 4, col = lane & 15, mk = col >> 1, reim = col & 1;
    const bool on = mk < nb;
    double Be[NR][16], Bo[NR][16];
    {
        const double* __restrict__ g0 = ph + (int64_t)(k0 + (on ? mk : 0)) * ph_stride + reim;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double* g = g0 + d_phidx(lmax + 1, pb[r] + kq + 4 * q, m);
                const double n = on ? g[0] : 0.0, s = on ? g[2] : 0.0;
                Be[r][q] = n + s;
                Bo[r][q] = n - s;
            }
    }
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    double* __restrict__ outp = part + chunk * part_chunk_stride + 2 * (mo - m) + (int64_t)(k0 + mk) * part_map_stride + reim;
    double* __restrict__ Tw = tile[wid];
    const double* __restrict__ To = tile[1 - wid];
    const int arow = (lane & 15) * 2 * kMxPitch + kq;
    const double* __restrict__ ph9 = ph + (int64_t)(k0 + 8) * ph_stride;     // X9 only
    double* __restrict__ out9 = part + chunk * part_chunk_stride + 2 * (mo - m) + (int64_t)(k0 + 8) * part_map_stride;
    double g9[NR][4];            // X9: (N+S).re, (N+S).im, (N-S).re, (N-S).im of pair 4 (lane & 15) + (lane >> 4)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        g9[r][0] = g9[r][1] = g9[r][2] = g9[r][3] = 0.0;
        if (X9) {
            const double* g = ph9 + d_phidx(lmax + 1, pb[r] + 4 * (lane & 15) + kq, m);
            g9[r][0] = g[0] + g[2];
            g9[r][1] = g[1] + g[3];
            g9[r][2] = g[0] - g[2];
            g9[r][3] = g[1] - g[3];
        }
    }
    for (int l0 = lw; l0 <= lmax; l0 += kMxL) {
        mx_d4 De0 = {0.0, 0.0, 0.0, 0.0}, De1 = De0, Do0 = De0, Do1 = De0;
        double xer = 0.0, xei = 0.0, xor_ = 0.0, xoi = 0.0;   // ninth map: even / odd rows, re / im; lane = (row, kq)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (l0 + kMxL <= lwr[r]) continue;            // no pair of this sub-block has started yet (wave-uniform)
            // seeds are injected only in the groups in which a lane of THIS block starts (with uniform block starts,
            // plan_tables.cpp, that is its first group alone)
            if (l0 <= lhi[r]) mx_recur<true>(al, l0, x[r], mc[r], mp[r], sc[r], sp[r], ls[r], Tw + lane);
            else              mx_recur<false>(al, l0, x[r], mc[r], mp[r], sc[r], sp[r], ls[r], Tw + lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define CMDR_MX_Q(q)                                                                                       \
            {                                                                                              \
                const double ae0 = Tw[arow + 4 * (q)], ao0 = Tw[arow + kMxPitch + 4 * (q)];                \
                const double ae1 = Tw[arow + 4 * (q) + 4], ao1 = Tw[arow + kMxPitch + 4 * (q) + 4];        \
                De0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ae0, Be[r][(q)], De0, 0, 0, 0);                 \
                Do0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ao0, Bo[r][(q)], Do0, 0, 0, 0);                 \
                De1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ae1, Be[r][(q) + 1], De1, 0, 0, 0);             \
                Do1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ao1, Bo[r][(q) + 1], Do1, 0, 0, 0);             \
                if (X9) {                                                                                  \
                    fmac_row_bcast<(q)>(xer, g9[r][0], ae0);                                               \
                    fmac_row_bcast<(q)>(xei, g9[r][1], ae0);                                               \
                    fmac_row_bcast<(q)>(xor_, g9[r][2], ao0);                                              \
                    fmac_row_bcast<(q)>(xoi, g9[r][3], ao0);                                               \
                    fmac_row_bcast<(q) + 1>(xer, g9[r][0], ae1);                                           \
                    fmac_row_bcast<(q) + 1>(xei, g9[r][1], ae1);                                           \
                    fmac_row_bcast<(q) + 1>(xor_, g9[r][2], ao1);                                          \
                    fmac_row_bcast<(q) + 1>(xoi, g9[r][3], ao1);                                           \
                }                                                                                          \
            }
            CMDR_MX_Q(0) CMDR_MX_Q(2) CMDR_MX_Q(4) CMDR_MX_Q(6) CMDR_MX_Q(8) CMDR_MX_Q(10) CMDR_MX_Q(12) CMDR_MX_Q(14)
#undef CMDR_MX_Q
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        De0 += De1;
        Do0 += Do1;
        // wave 0 finishes the even rows (l - l0 even), wave 1 the odd ones: hand the other half over
        const mx_d4 give = wid == 0 ? Do0 : De0;
        mx_d4 keep = wid == 0 ? De0 : Do0;
#pragma unroll
        for (int v = 0; v < 4; ++v) Tw[v * 64 + lane] = give[v];
        double k9r = 0.0, k9i = 0.0;
        if (X9) {   // rows summed over the four pair lanes (kq); then the same hand-over as the matrix-unit rows
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                xer += __shfl_xor(xer, o);
                xei += __shfl_xor(xei, o);
                xor_ += __shfl_xor(xor_, o);
                xoi += __shfl_xor(xoi, o);
            }
            k9r = wid == 0 ? xer : xor_;
            k9i = wid == 0 ? xei : xoi;
            if (lane < 16) {
                Tw[256 + lane] = wid == 0 ? xor_ : xer;
                Tw[272 + lane] = wid == 0 ? xoi : xei;
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v) keep[v] += To[v * 64 + lane];
        if (on) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int l = l0 + 2 * (kq + 4 * v) + wid;
                if (l <= lmax) outp[2 * l] = keep[v];
            }
        }
        if (X9 && lane < 16) {
            const int l = l0 + 2 * lane + wid;
            if (l <= lmax) {
                out9[2 * l] = k9r + To[256 + lane];
                out9[2 * l + 1] = k9i + To[272 + lane];
            }
        }
        __syncthreads();
    }
}

// ---- adjoint of 1..7 maps without the matrix unit: the same task shape, recursion and LDS transposition as k_leg_adj_mx,
// and for EVERY map the accumulation k_leg_adj_mx<.., X9> uses for its ninth: lane = (l row, pair kq of quad q), the
// map's N+S / N-S phases in 4 registers per 64-pair block with lane (row q', kq) holding pair 4 q' + kq, four
// v_fmac_f64_dpp row_newbcast:q per map and step.  2 + 2 NB fp64 operations per (ring pair, l) -- what the synthesis
// needs -- against ~4.8 per map in k_leg_adj (wave-wide reductions) and a fixed 8-map price in k_leg_adj_mx.  The four
// pair lanes of a row are folded with two half / row swaps per map (swap_halves, swap_rows), which also packs a map's
// four sums into one register: row 0 = even-l re, 1 = odd-l re, 2 = even-l im, 3 = odd-l im.
template <int NR, int NB>
__global__ void __launch_bounds__(128) k_leg_adj_dx(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                    const double* __restrict__ ph, int64_t ph_stride, int k0,
                                                    double* __restrict__ part, int64_t part_map_stride,
                                                    int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double tile[2][kMxL * kMxPitch];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= ntasks) return;
    const WaveTask T = tasks[blockIdx.x];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lmax = A.lmax;
    const int kq = lane >> 4, row = lane & 15;
    int pb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sub = NR == 1 ? wid : (wid == 0 ? 3 * r : 1 + r);
        pb[r] = chunk * (128 * NR) + sub * 64;
    }
    double x[NR], mc[NR], mp[NR], sc[NR], sp[NR];
    int ls[NR], lwr[NR], lhi[NR];
    double g[NR][NB][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = pb[r] + lane;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x[r] = A.x[p];
        ls[r] = A.ls[idx];
        sc[r] = A.seedc[idx];
        sp[r] = A.seedp[idx];
        mc[r] = mp[r] = 0.0;
        int v = ls[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        lwr[r] = __builtin_amdgcn_readfirstlane(v);
        v = ls[r] == 0x3fffffff ? -1 : ls[r];               // last start among the lanes that start at all
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        lhi[r] = __builtin_amdgcn_readfirstlane(v);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const double* gp = ph + (int64_t)(k0 + k) * ph_stride + d_phidx(lmax + 1, pb[r] + 4 * row + kq, m);
            g[r][k][0] = gp[0] + gp[2];
            g[r][k][1] = gp[1] + gp[3];
            g[r][k][2] = gp[0] - gp[2];
            g[r][k][3] = gp[1] - gp[3];
        }
    }
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    // this lane's output slot: row pair (kq & 1) = parity of l, (kq >> 1) = re / im
    double* __restrict__ outp = part + chunk * part_chunk_stride + 2 * (mo - m) + (int64_t)k0 * part_map_stride + (kq >> 1);
    double* __restrict__ Tw = tile[wid];
    const double* __restrict__ To = tile[1 - wid];
    const int arow = row * 2 * kMxPitch + kq;
    for (int l0 = lw; l0 <= lmax; l0 += kMxL) {
        double acc[NB][4];
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (l0 + kMxL <= lwr[r]) continue;
            // seeds are injected only in the groups in which a lane of THIS block starts (with uniform block starts,
            // plan_tables.cpp, that is its first group alone)
            if (l0 <= lhi[r]) mx_recur<true>(al, l0, x[r], mc[r], mp[r], sc[r], sp[r], ls[r], Tw + lane);
            else              mx_recur<false>(al, l0, x[r], mc[r], mp[r], sc[r], sp[r], ls[r], Tw + lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define CMDR_DX_Q(q)                                                                                       \
            {                                                                                              \
                const double ae = Tw[arow + 4 * (q)], ao = Tw[arow + kMxPitch + 4 * (q)];                  \
                _Pragma("unroll") for (int k = 0; k < NB; ++k) {                                           \
                    fmac_row_bcast<(q)>(acc[k][0], g[r][k][0], ae);                                        \
                    fmac_row_bcast<(q)>(acc[k][1], g[r][k][1], ae);                                        \
                    fmac_row_bcast<(q)>(acc[k][2], g[r][k][2], ao);                                        \
                    fmac_row_bcast<(q)>(acc[k][3], g[r][k][3], ao);                                        \
                }                                                                                          \
            }
            CMDR_DX_Q(0) CMDR_DX_Q(1) CMDR_DX_Q(2) CMDR_DX_Q(3) CMDR_DX_Q(4) CMDR_DX_Q(5) CMDR_DX_Q(6) CMDR_DX_Q(7)
            CMDR_DX_Q(8) CMDR_DX_Q(9) CMDR_DX_Q(10) CMDR_DX_Q(11) CMDR_DX_Q(12) CMDR_DX_Q(13) CMDR_DX_Q(14) CMDR_DX_Q(15)
#undef CMDR_DX_Q
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // fold the four pair lanes (kq) of every row; t[k]: 16-lane row 0 = even-l re, 1 = odd-l re, 2 = even-l im, 3 = odd-l im
        double t[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            double a0 = acc[k][0], a1 = acc[k][1], a2 = acc[k][2], a3 = acc[k][3];
            swap_halves(a0, a1);
            swap_halves(a2, a3);
            double s01 = a0 + a1, s23 = a2 + a3;     // rows: s01 = e_re(0+2), e_re(1+3), e_im(0+2), e_im(1+3); s23 likewise odd
            swap_rows(s01, s23);
            t[k] = s01 + s23;
            Tw[k * 64 + lane] = t[k];
        }
        __syncthreads();
        // wave 0 writes the even-l rows (0, 2), wave 1 the odd-l ones (1, 3): this wave's sub-blocks + the other wave's
        const int l = l0 + 2 * row + (kq & 1);
        if ((kq & 1) == wid && l <= lmax) {
#pragma unroll
            for (int k = 0; k < NB; ++k) outp[(int64_t)k * part_map_stride + 2 * l] = t[k] + To[k * 64 + lane];
        }
        __syncthreads();
    }
}

// ---- Synthesis on the matrix unit, small tile (round 3) ------------------------------------------------------------
// v_mfma_f64_4x4x4_4b_f64 -- four independent 4x4x4 blocks per instruction -- sustains 72 TFLOP/s on this part where the
// 16x16x4 tile sustains 49 and a loop of fp64 VALU FMAs 65 (tools/microbench/fp64_mfma_4x4.hip; one shared datapath).
// Operand layout, found by experiment (tools/microbench/mfma_4x4_layout.hip):  lane = x + 4 b + 16 y,
//     A_b[i][k]: x = i, y = k      B_b[k][j]: x = j, y = k      D_b[i][j]: x = j, y = i      (b = block; CBSZ/ABID: no effect)
// The synthesis F(pair, col) = sum_l mu_l(pair) a_l(col) maps onto it with block = 4 ring pairs, i = pair of the block,
// k = 4 values of l of one parity, j = 4 columns (two maps x (re, im)): the 16 lanes of a row (lane & 15) are 16 ring
// pairs in A and D, every output belongs to one lane (no partial sums), and the coefficients -- the same for every
// block -- are read from the workgroup's LDS tile with a broadcast address.  Per (ring pair, l): 2 fp64 operations of
// recursion + 8 flop per column group, all maps of a batch on ONE recursion (nine maps: five groups, where
// k_leg_synth_wg runs 5 + 4 maps on two); one wave = one 64-pair block (the accumulators of all l stay in 8 NCG
// registers), a workgroup = the four blocks of two adjacent tasks of one m.
//   * mu goes from the recursion (lane = ring pair) to the A operand (lane = (pair of 16, l of 4)) through a
//     wave-private LDS image of 16 l x 64 pairs, two of them: while the 16 NCG MFMAs of one image are issued, the 16
//     recursion steps that fill the other are slotted between them (one step per NCG MFMAs), so neither the dependent
//     FMA chain nor the LDS round trip is ever waited for.  Pitch 66 doubles and operand rows {r, r + 8, r + 2, r + 10}:
//     the two half-waves of a ds_read_b64 touch 32 different banks.
//   * the recursion runs in the sign-alternated form nu_l = s_l mu_l, s = (+, +, -, -) from l = m on:
//     nu_{l+1} = nu_{l-1} + (-1)^(l-m) alpha_{l+1} (x nu_l) is ONE multiply and ONE v_fmac_f64_dpp into the register
//     that held nu_{l-1}, with the 16 alphas of an image in one VGPR (lane j of every row: one vector load per image,
//     requested an image ahead) and the step's alpha delivered by the DPP row broadcast -- no scalar loads, whose waits
//     would stall the MFMA stream.  The signs s_l are folded into the coefficient tile when it is staged.
//   * the l grid of the images is anchored at m, so the order of every sum depends on (m, pair) alone.
struct PrepTermM4 { long long sxo; int wo, lmaxc; };
constexpr int kM4H = 16;        // l per mu image
constexpr int kM4Pitch = 66;    // doubles per image row
template <int NCG>
constexpr int m4_lds_bytes() {
    return (int)sizeof(double) * (4 * 2 * kM4H * kM4Pitch + 2 * kTileL * (4 * NCG + 1)) + (int)sizeof(PrepTermM4) * 2 * NCG * 8;
}

// One image: MF -- the 16 NCG MFMAs on image `ard` and coefficient rows `brd`; REC -- the 16 recursion steps of the NEXT
// image (l = l0n .. l0n + 15) into `twr`, step S behind the MFMAs of (operand group S / 4, pair group S % 4).
template <int NCG, bool MF, bool REC, bool INJ, int ROW, int S>
__device__ __forceinline__ void m4_image(const double* __restrict__ ard, const double* __restrict__ brd,
                                         double (&acc)[4][NCG][2], double (&Ac)[4], double (&Bc)[NCG], double (&An)[4],
                                         double (&Bn)[NCG], double* __restrict__ twr, int l0n, double x, double& na,
                                         double& nb, double scn, double spn, int ls, double cv) {
    if constexpr (S < 16) {
        constexpr int g = S >> 2, q = S & 3;
        if (MF && q == 0 && g < 3) {          // operands of group g + 1: requested before the MFMAs of group g are issued
            constexpr int r1 = 4 * ((g + 1) >> 1) + ((g + 1) & 1);
#pragma unroll
            for (int c = 0; c < NCG; ++c) Bn[c] = brd[r1 * ROW + 4 * c];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) An[qq] = ard[r1 * kM4Pitch + 16 * qq];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MF) {
#pragma unroll
            for (int c = 0; c < NCG; ++c)
                acc[q][c][g & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ac[q], Bc[c], acc[q][c][g & 1], 0, 0, 0);
        }
        if (REC) {
            double& cur = (S & 1) ? nb : na;
            double& prev = (S & 1) ? na : nb;
            if (INJ) if (ls == l0n + S) { cur = scn; prev = spn; }
            twr[S * kM4Pitch] = cur;
            const double xm = x * cur;
            fmac_row_bcast<S>(prev, cv, xm);      // prev <- nu_{l+1}
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MF && q == 3 && g < 3) {
#pragma unroll
            for (int c = 0; c < NCG; ++c) Bc[c] = Bn[c];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) Ac[qq] = An[qq];
        }
        m4_image<NCG, MF, REC, INJ, ROW, S + 1>(ard, brd, acc, Ac, Bc, An, Bn, twr, l0n, x, na, nb, scn, spn, ls, cv);
    }
}

template <int NCG, bool PREP>
__global__ void __launch_bounds__(256, 2) k_leg_synth_m4(LegArgs A, const WaveTask* __restrict__ tasks, int ntasks,
                                                         const double* __restrict__ ast, int nbs, int k0, int nb,
                                                         double* __restrict__ ph, int64_t ph_stride, PrepDev P) {
    constexpr int NS = 2 * NCG;                          // map slots (slots >= nb stay zero and are not stored)
    constexpr int ROW = 2 * NS + 1;                      // odd: rows 8 apart fall into different banks
    constexpr int NE = kTileL * ROW;
    constexpr int NLD = (NE + 255) / 256;
    static_assert(kTileL == 2 * kM4H, "two mu images per coefficient tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_m4[];
    double* const sm = reinterpret_cast<double*>(smem_m4);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    double* const Tw = sm + wid * (2 * kM4H * kM4Pitch);          // two images
    double* const ctile = sm + 4 * (2 * kM4H * kM4Pitch);
    PrepTermM4* const terms = reinterpret_cast<PrepTermM4*>(ctile + 2 * NE);
    const int tg = blockIdx.x >> 1, half = blockIdx.x & 1;
    const WaveTask T = tasks[tg * 4 + half * 2 + (wid >> 1)];
    const int m = __builtin_amdgcn_readfirstlane(tasks[tg * 4].m);
    const int lw0 = __builtin_amdgcn_readfirstlane(min(tasks[tg * 4 + half * 2].lw, tasks[tg * 4 + half * 2 + 1].lw));
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lmax = A.lmax;
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ as = ast + 2 * ((int64_t)nbs * (mo - m) + k0);
    const int64_t ls2 = 2 * (int64_t)nbs;
    const int pbase = (chunk < 0 ? 0 : chunk) * 128 + (wid & 1) * 64;
    double x, na = 0.0, nb_ = 0.0, scn, spn;
    int ls, lwr;
    {
        const int p = pbase + lane;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x = A.x[p];
        ls = chunk < 0 ? 0x3fffffff : A.ls[idx];
        const double sc = A.seedc[idx], sp = A.seedp[idx];
        scn = ((ls - m) & 2) ? -sc : sc;                 // seeds in the sign-alternated form
        spn = ((ls - 1 - m) & 2) ? -sp : sp;
        int v = ls;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        lwr = __builtin_amdgcn_readfirstlane(v);
    }
    double acc[4][NCG][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < NCG; ++c) acc[q][c][0] = acc[q][c][1] = 0.0;
    // ---- coefficient tiles: as k_leg_synth_wg (PREP: k_band_prep folded into the staging), rows with (l - m) & 2 negated
    const int sl = m == 0 ? 1 : 2;
    const int64_t na_ = (int64_t)(lmax + 1) * (lmax + 1);
    const int64_t gbase = d_packed_index(lmax, 0, m);
    if (PREP) {
        if ((int)threadIdx.x < nb * P.ncomp) {
            const int k = threadIdx.x / P.ncomp, c = threadIdx.x - k * P.ncomp, bm = k0 + k;
            const CompDev C = P.comps[c];
            const int st = P.bm_stokes[bm];
            PrepTermM4 Tm;
            Tm.lmaxc = st < C.nmaps ? C.lmax : -1;
            Tm.sxo = C.pos + (int64_t)st * C.nalm + d_packed_index(C.lmax, 0, m);
            Tm.wo = (bm * P.ncomp + c) * (lmax + 1);
            terms[threadIdx.x] = Tm;
        }
        __syncthreads();
    }
    const bool prep1 = PREP && P.ncomp == 1 && !P.extra;
    const double* pw[NLD];
    const double* psx[NLD];
    int plm[NLD], prow[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        const int row = e / ROW, col = e - row * ROW;
        prow[i] = row;
        pw[i] = psx[i] = nullptr;
        plm[i] = -1;
        if (PREP && prep1 && e < NE && col < 2 * nb) {
            const PrepTermM4 Tm = terms[col >> 1];
            pw[i] = P.w + Tm.wo;
            psx[i] = P.sx + Tm.sxo + (col & 1);
            plm[i] = ((col & 1) && m == 0) ? -1 : min(Tm.lmaxc, lmax);
        }
    }
    const double* __restrict__ pcn = PREP ? P.cnorm + (mo - m) : nullptr;
    const double kap = m == 0 ? 1.0 : 0.70710678118654752440;
    // The loads of tile t + 1 are issued at the top of tile t and their arithmetic is done at its bottom (fetch_finish):
    // done right behind the loads it would put a vmcnt(0) wait -- a full trip to L2 / HBM -- in front of every tile's MFMAs
    double raw[NLD][3];
    auto fetch = [&](int lb, double* v) {
        if (PREP && prep1) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int l = lb + prow[i];
                raw[i][0] = raw[i][1] = raw[i][2] = 0.0;
                if (l <= plm[i]) {
                    raw[i][0] = pw[i][l];
                    raw[i][1] = psx[i][sl * l];
                    raw[i][2] = pcn[l];
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = (int)threadIdx.x + 256 * i;
            const int row = e / ROW, col = e - row * ROW;
            const int l = lb + row;
            double val = 0.0;
            if (e < NE && l <= lmax && col < 2 * nb) {
                if (PREP) {
                    const int k = col >> 1, part = col & 1;
                    if (!(part && m == 0)) {
                        double s = 0.0;
                        if (P.extra) s = P.extra[(int64_t)(k0 + k) * na_ + gbase + sl * l + part];
                        for (int c = 0; c < P.ncomp; ++c) {
                            const PrepTermM4 Tm = terms[k * P.ncomp + c];
                            if (l > Tm.lmaxc) continue;
                            const double wc = P.w[Tm.wo + l];
                            const double t = P.sx[Tm.sxo + sl * l + part];
                            if (wc != 0.0) s += wc * t;
                        }
                        val = s * (P.cnorm[mo - m + l] * kap);
                    }
                } else {
                    val = as[ls2 * l + col];
                }
            }
            v[i] = (row & 2) ? -val : val;                // s_l of the sign-alternated recursion (lb == m mod 32)
        }
    };
    auto fetch_finish = [&](double* v) {
        if (PREP && prep1) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                double a = 0.0;
                if (raw[i][0] != 0.0) a += raw[i][0] * raw[i][1];
                a *= raw[i][2] * kap;
                v[i] = (prow[i] & 2) ? -a : a;
            }
        }
    };
    // operand addressing: k = lane >> 4 picks the row offset {0, 8, 2, 10}; A reads pair (lane & 15) of a 16-pair
    // group, B column (lane & 3) of a column group
    const int kk = lane >> 4, roff = (kk & 1) * 8 + (kk >> 1) * 2;
    const int aoff = roff * kM4Pitch + (lane & 15);
    const int boff = roff * ROW + (lane & 3);
    // the 16 signed alphas of the image that starts at l0: lane j of every row holds (-1)^j alpha_{l0 + j + 1}
    auto load_al = [&](int l0) {
        return al[min(l0 + 1 + (lane & 15), lmax + 2)];     // (behind lmax: values nobody uses, reads inside the table)
    };
    auto signed_al = [&](double a) { return (lane & 1) ? -a : a; };
    const int lb0 = lw0 - ((lw0 - m) & (kTileL - 1));
    const int ntile = lb0 > lmax ? 0 : (lmax - lb0) / kTileL + 1;
    double pre[NLD];
    if (ntile > 0) { fetch(lb0, pre); fetch_finish(pre); }
    bool primed = false;                                  // wave-uniform: image 0 of the current tile is filled
    double cv[2] = {0.0, 0.0}, cvr[2] = {0.0, 0.0};   // signed alphas beside image 0 / 1 of this tile; raw loads for the next
    double Ac[4], Bc[NCG], An[4], Bn[NCG];
    for (int t = 0; t < ntile; ++t) {
        double* cur = ctile + (t & 1) * NE;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = (int)threadIdx.x + 256 * i;
            if (e < NE) cur[e] = pre[i];
        }
        __syncthreads();                                  // tile t visible; tile t-1 (same buffer as t+1) fully consumed
        const int lb = lb0 + t * kTileL;
        const bool active = chunk >= 0 && lb + kTileL > lwr;   // some pair of this block has started (wave-uniform)
        // the alphas of the two recursions that run beside the NEXT tile's images: requested first, so that waiting for
        // them at the bottom of this tile does not wait for the coefficient loads behind them
        if (active) { cvr[0] = load_al(lb + kTileL + kM4H); cvr[1] = load_al(lb + 2 * kTileL); }
        if (t + 1 < ntile) fetch(lb + kTileL, pre);       // in flight while this tile is consumed
        if (!active) {
            if (t + 1 < ntile) fetch_finish(pre);
            continue;
        }
        if (!primed) {                                    // first image of this wave: recursion alone
            const double c0 = signed_al(load_al(lb));
            m4_image<NCG, false, true, true, ROW, 0>(nullptr, nullptr, acc, Ac, Bc, An, Bn, Tw + lane, lb, x, na, nb_, scn, spn, ls, c0);
            cv[0] = signed_al(load_al(lb + kM4H));
            cv[1] = signed_al(load_al(lb + kTileL));
            primed = true;
        }
        // the two images of the tile: MFMAs on image h, the recursion of the next image (the other buffer) beside them.
        // Behind the last tile that recursion runs on into rows nobody reads (one code path: a second instantiation
        // of the image body costs the register allocator its grip on the 8 NCG accumulators)
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double* ard = Tw + h * (kM4H * kM4Pitch) + aoff;
            const double* brd = cur + h * (kM4H * ROW) + boff;
            const int l0n = lb + (h + 1) * kM4H;
#pragma unroll
            for (int c = 0; c < NCG; ++c) Bc[c] = brd[4 * c];
#pragma unroll
            for (int q = 0; q < 4; ++q) Ac[q] = ard[16 * q];
#ifndef CMDR_M4_DBG
#define CMDR_M4_DBG 0      // timing experiments (separate builds): 1 no recursion beside the MFMAs, 2 no MFMAs
#endif
            m4_image<NCG, CMDR_M4_DBG != 2, CMDR_M4_DBG != 1, true, ROW, 0>(ard, brd, acc, Ac, Bc, An, Bn, Tw + (1 - h) * (kM4H * kM4Pitch) + lane, l0n, x,
                                                    na, nb_, scn, spn, ls, h == 0 ? cv[0] : cv[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        cv[0] = signed_al(cvr[0]);
        cv[1] = signed_al(cvr[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntile) fetch_finish(pre);
    }
    if (chunk < 0) return;
    // D_b[i][j]: lane = j + 4 b + 16 i holds pair 16 q + 4 b + i, column j of group c
    const int j = lane & 3, slot0 = j >> 1, reim = j & 1;
    const int pl = pbase + 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
    for (int c = 0; c < NCG; ++c) {
        const int slot = 2 * c + slot0;
        if (slot >= nb) continue;
        double* __restrict__ o0 = ph + (int64_t)(k0 + slot) * ph_stride + reim;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double* o = o0 + d_phidx(lmax + 1, pl + 16 * q, m);
            o[0] = acc[q][c][0] + acc[q][c][1];
            o[2] = acc[q][c][0] - acc[q][c][1];
        }
    }
}

// maps sharing one recursion per wave (register budget).  Tuning knobs: CMDR_LEG_NB caps both kernels,
// CMDR_LEG_NB_S / CMDR_LEG_NB_A set the synthesis / adjoint value (up to the compiled maximum).
static int leg_batch(int R, bool adjoint, bool wg = false) {
    // R = 4: 3 measured best for both kernels (9 maps: 3+3+3); synthesis in the workgroup form at R = 2: 5 (9 = 5+4)
    const int cap = R == 1 ? 9 : (R == 2 ? (wg && !adjoint ? 5 : 4) : (adjoint ? 3 : 4));
    int nb = R == 1 ? 9 : (R == 2 ? (wg && !adjoint ? 5 : 4) : 3);
    if (const char* e = std::getenv(adjoint ? "CMDR_LEG_NB_A" : "CMDR_LEG_NB_S")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= cap) nb = v;
    }
    if (const char* e = std::getenv("CMDR_LEG_NB")) {
        const int v = std::atoi(e);
        if (v >= 1 && v < nb) nb = v;
    }
    return nb;
}
int leg_max_batch(int R) { return leg_batch(R, false); }

template <int R, int NB>
static void synth_RN(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, int nbs, int k0, int rep,
                     double* ph, int64_t ph_stride, hipStream_t s, const PrepDev* prep) {
    const bool dppc = [] { const char* e = std::getenv("CMDR_SYNTH_DPP"); return !e || std::atoi(e) != 0; }();   // per call (test hook)
    constexpr int NBW = NB <= 5 ? NB : 5;
    const dim3 grid((ntasks / 4) * rep);
#define CMDR_WG(PP, DD, UU) hipLaunchKernelGGL((k_leg_synth_wg<R, NBW, PP, DD, UU>), grid, dim3(256), 0, s, A, tasks, ntasks, ast, \
                                               nbs, k0, rep, ph, ph_stride, prep ? *prep : PrepDev{})
    if (A.wg && NB <= 5) {
        const bool uni = dppc && A.uni;
        if (prep) { if (uni) CMDR_WG(true, true, true); else if (dppc) CMDR_WG(true, true, false); else CMDR_WG(true, false, false); }
        else      { if (uni) CMDR_WG(false, true, true); else if (dppc) CMDR_WG(false, true, false); else CMDR_WG(false, false, false); }
    } else {
        hipLaunchKernelGGL((k_leg_synth<R, NB>), grid, dim3(256), 0, s, A, tasks, ntasks, ast, nbs, k0, rep, ph, ph_stride);
    }
#undef CMDR_WG
}
// balanced split of nmaps into batches of at most nbmax maps, e.g. 9 -> 3+3+3, 8 -> 3+3+2; consecutive batches of
// equal size share one launch: calls f(nb, k0, rep)
template <class F>
static void for_batches(int nmaps, int nbmax, F f) {
    const int nbatch = (nmaps + nbmax - 1) / nbmax;
    int k0 = 0, ib = 0;
    while (ib < nbatch) {
        const int nb = (nmaps - k0 + (nbatch - ib) - 1) / (nbatch - ib);
        int rep = 1, k = k0 + nb;
        while (ib + rep < nbatch && (nmaps - k + (nbatch - ib - rep) - 1) / (nbatch - ib - rep) == nb) { ++rep; k += nb; }
        f(nb, k0, rep);
        k0 = k;
        ib += rep;
    }
}
bool leg_synth_can_prep(const LegArgs& A) {
    const bool on = [] { const char* e = std::getenv("CMDR_SYNTH_PREP"); return !e || std::atoi(e) != 0; }();   // per call (test hook)
    return on && A.wg;
}
void launch_leg_synth(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, double* ph,
                      int64_t ph_stride, int nmaps, hipStream_t s, int nbs, const PrepDev* prep) {
    if (ntasks == 0 || nmaps == 0) return;
    if (nbs < 0) nbs = nmaps;
    // CMDR_SYNTH_M4=n: batches of n maps and more through the small-tile matrix-unit kernel, up to ten maps on one
    // recursion.  OFF by default (0): measured 3.3-3.9 ms against 2.7 ms for nine maps at Nside 1024 / lmax 2000 -- the
    // 512 B a wave stores per l for the transposition meet the 85 B/clk of the CU's LDS store path, and the recursion's
    // VALU instructions do not hide behind the same wave's MFMAs (tried list).  Read per call (test hook).
    const int m4_min = [] { const char* e = std::getenv("CMDR_SYNTH_M4"); return e ? std::atoi(e) : 0; }();
    if (A.wg && A.R == 2 && m4_min > 0 && nmaps >= m4_min) {
        for_batches(nmaps, 10, [&](int nb, int k0, int rep) {
            for (int ir = 0; ir < rep; ++ir, k0 += nb) {
#define CMDR_M4(NCG)                                                                                                    \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_leg_synth_m4<NCG, true>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, m4_lds_bytes<NCG>());                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_leg_synth_m4<NCG, false>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, m4_lds_bytes<NCG>());                 \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        if (prep)                                                                                                       \
            hipLaunchKernelGGL((k_leg_synth_m4<NCG, true>), dim3((ntasks / 4) * 2), dim3(256), m4_lds_bytes<NCG>(), s,  \
                               A, tasks, ntasks, ast, nbs, k0, nb, ph, ph_stride, *prep);                               \
        else                                                                                                            \
            hipLaunchKernelGGL((k_leg_synth_m4<NCG, false>), dim3((ntasks / 4) * 2), dim3(256), m4_lds_bytes<NCG>(), s, \
                               A, tasks, ntasks, ast, nbs, k0, nb, ph, ph_stride, PrepDev{});                           \
    } while (0)
                switch ((nb + 1) / 2) {
                    case 1: CMDR_M4(1); break;
                    case 2: CMDR_M4(2); break;
                    case 3: CMDR_M4(3); break;
                    case 4: CMDR_M4(4); break;
                    default: CMDR_M4(5); break;
                }
#undef CMDR_M4
            }
        });
        return;
    }
    for_batches(nmaps, A.wg ? std::min(leg_batch(A.R, false, true), 5) : leg_batch(A.R, false), [&](int nb, int k0, int rep) {
#define CMDR_S(RR, NN) case NN: synth_RN<RR, NN>(A, tasks, ntasks, ast, nbs, k0, rep, ph, ph_stride, s, prep); break;
        if (A.R == 1) {
            switch (nb) { CMDR_S(1, 1) CMDR_S(1, 2) CMDR_S(1, 3) CMDR_S(1, 4) CMDR_S(1, 5) CMDR_S(1, 6) CMDR_S(1, 7)
                          CMDR_S(1, 8) CMDR_S(1, 9) }
        } else if (A.R == 2) {
            switch (nb) { CMDR_S(2, 1) CMDR_S(2, 2) CMDR_S(2, 3) CMDR_S(2, 4) CMDR_S(2, 5) }
        } else {
            switch (nb) { CMDR_S(4, 1) CMDR_S(4, 2) CMDR_S(4, 3) CMDR_S(4, 4) }
        }
#undef CMDR_S
    });
}

template <int R, int NB, bool SQ>
static void adj_RN(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride, int k0,
                   int rep, double* part, int64_t pms, int64_t pcs, hipStream_t s) {
    hipLaunchKernelGGL((k_leg_adj<R, NB, SQ>), dim3((ntasks / 4) * rep), dim3(256), 0, s, A, tasks, ntasks, ph,
                       ph_stride, k0, rep, part, pms, pcs);
}
void launch_leg_adj(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                    double* part, int64_t pms, int64_t pcs, int nmaps, bool square, hipStream_t s,
                    const std::function<void(int)>& between) {
    if (ntasks == 0 || nmaps == 0) return;
    if (square) {  // setup-time only (noise diagonal): one map at a time
        for (int k0 = 0; k0 < nmaps; ++k0) {
            if (A.R == 1) adj_RN<1, 1, true>(A, tasks, ntasks, ph, ph_stride, k0, 1, part, pms, pcs, s);
            else if (A.R == 2) adj_RN<2, 1, true>(A, tasks, ntasks, ph, ph_stride, k0, 1, part, pms, pcs, s);
            else adj_RN<4, 1, true>(A, tasks, ntasks, ph, ph_stride, k0, 1, part, pms, pcs, s);
        }
        return;
    }
    // 6..8 maps (9 with the last one riding along) per launch on the matrix unit; 3..5 maps through the DPP form of the
    // same task (k_leg_adj_dx: 1.25 + 0.23 nb ms at the cfg3 geometry against a flat 2.47 ms); 1..2 maps, plans with one
    // ring pair per lane and `square` through the VALU kernel k_leg_adj (0.93 ms per map)
    const int mx_min = [] { const char* e = std::getenv("CMDR_ADJ_MX"); return e ? std::atoi(e) : 6; }();   // read per call (test hook)
    const bool x9_on = [] { const char* e = std::getenv("CMDR_ADJ_X9"); return !e || std::atoi(e) != 0; }();
    const bool dx_on = [] { const char* e = std::getenv("CMDR_ADJ_DX"); return !e || std::atoi(e) != 0; }();
    int kdone = 0, nmx = 0;
    if ((A.R == 4 || A.R == 2) && mx_min > 0)
        for (int left = nmaps; left >= mx_min; left -= std::min(8, left)) nmx += std::min(8, left);
    while (kdone < nmx) {
        const int nb = std::min(8, nmx - kdone);
        const bool x9 = x9_on && nb == 8 && nmaps - (kdone + 8) == 1;    // a single map left over rides along
#define CMDR_MX(NRR, XX) hipLaunchKernelGGL((k_leg_adj_mx<NRR, XX>), dim3(ntasks), dim3(128), 0, s, A, tasks, ntasks, ph, \
                                            ph_stride, kdone, nb, part, pms, pcs)
        if (A.R == 4) { if (x9) CMDR_MX(2, true); else CMDR_MX(2, false); }
        else          { if (x9) CMDR_MX(1, true); else CMDR_MX(1, false); }
#undef CMDR_MX
        kdone += nb + (x9 ? 1 : 0);
    }
    if (between) between(kdone);
    if (kdone == nmaps) return;
    if (dx_on && (A.R == 4 || A.R == 2) && nmaps - kdone >= 3) {
        const int nb = nmaps - kdone;       // 3..5 (6 and more went to the matrix unit), or up to 7 with CMDR_ADJ_MX raised
#define CMDR_DX(NN) case NN:                                                                                              \
        if (A.R == 4) hipLaunchKernelGGL((k_leg_adj_dx<2, NN>), dim3(ntasks), dim3(128), 0, s, A, tasks, ntasks, ph,      \
                                         ph_stride, kdone, part, pms, pcs);                                               \
        else          hipLaunchKernelGGL((k_leg_adj_dx<1, NN>), dim3(ntasks), dim3(128), 0, s, A, tasks, ntasks, ph,      \
                                         ph_stride, kdone, part, pms, pcs);                                               \
        return;
        switch (nb) { CMDR_DX(3) CMDR_DX(4) CMDR_DX(5) CMDR_DX(6) CMDR_DX(7) default: break; }
#undef CMDR_DX
    }
    ph += (int64_t)kdone * ph_stride;
    part += (int64_t)kdone * pms;
    nmaps -= kdone;
    for_batches(nmaps, leg_batch(A.R, true), [&](int nb, int k0, int rep) {
#define CMDR_A(RR, NN) case NN: adj_RN<RR, NN, false>(A, tasks, ntasks, ph, ph_stride, k0, rep, part, pms, pcs, s); break;
        if (A.R == 1) {
            switch (nb) { CMDR_A(1, 1) CMDR_A(1, 2) CMDR_A(1, 3) CMDR_A(1, 4) CMDR_A(1, 5) CMDR_A(1, 6) CMDR_A(1, 7)
                          CMDR_A(1, 8) CMDR_A(1, 9) }
        } else if (A.R == 2) {
            switch (nb) { CMDR_A(2, 1) CMDR_A(2, 2) CMDR_A(2, 3) CMDR_A(2, 4) }
        } else {
            switch (nb) { CMDR_A(4, 1) CMDR_A(4, 2) CMDR_A(4, 3) }
        }
#undef CMDR_A
    });
}

// ---- spin-2 (Q,U <-> E,B): same task / wave structure; one polarisation pair per launch slice
template <int R>
__global__ void __launch_bounds__(256) k_leg2_synth(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                    const double* __restrict__ st, int npol, int ip,
                                                    double* __restrict__ ph, int64_t ph_stride, int kq) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    leg2_synth_lane<R>(A, st, npol, ip, ph, ph_stride, kq, __builtin_amdgcn_readfirstlane(T.m),
                       __builtin_amdgcn_readfirstlane(T.chunk), __builtin_amdgcn_readfirstlane(T.lw),
                       __builtin_amdgcn_readfirstlane(T.lAend), threadIdx.x & 63);
}

template <int R>
__global__ void __launch_bounds__(256) k_leg2_adj(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                  const double* __restrict__ ph, int64_t ph_stride, int kq,
                                                  double* __restrict__ part, int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double lds[4][2 * kRedTile];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lmax = A.lmax;
    Leg2State<R> S;
    Adj2G<R> G;
    leg2_load_state<R>(A, m, chunk, lane, S);
    leg2_adj_load<R>(A, ph, ph_stride, kq, m, chunk, lane, G);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (S.ls[r] == lw) { S.pc[r] = S.sd(r, 0); S.pp[r] = S.sd(r, 1); S.mc[r] = S.sd(r, 2); S.mp[r] = S.sd(r, 3); }
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    double* __restrict__ out = part + chunk * part_chunk_stride + 4 * (mo - m);   // 4 doubles per l
    double* wl = lds[wid];
    const int id = lane >> 2;              // value this lane ends up with: l = l0 + id/4, component id & 3
    int buf = 0;
    double a4[4], b4[4];                   // (alpha, beta)_{l0 + 1 + j} of the current group (sload_d4: see k_leg_adj)
    {
        sgpr8 a = sload_d4(al + lw + 1), b = sload_d4(be + lw + 1);
        double dummy = 0.0;
        sload_wait(a, b, dummy);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a4[j] = sgpr_double(a, j); b4[j] = sgpr_double(b, j); }
    }
    for (int l0 = lw; l0 <= lmax; l0 += 4) {
        sgpr8 an = sload_d4(al + l0 + 5), bn = sload_d4(be + l0 + 5);
        double v[16];
        if (l0 < lAend) leg2_adj_group_v<R, true>(a4, b4, l0, S, G, v);
        else            leg2_adj_group_v<R, false>(a4, b4, l0, S, G, v);
        double sacc = wave_reduce16(v, wl + buf * kRedTile, lane);
        buf ^= 1;
        const int l = l0 + (id >> 2);
        if ((lane & 3) == 0 && l <= lmax) out[4 * l + (id & 3)] = sacc;
        sload_wait(an, bn, sacc);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a4[j] = sgpr_double(an, j); b4[j] = sgpr_double(bn, j); }
    }
}

// ---- two polarisation pairs per wave: the two spin-weighted recursions (and W, X) are shared, only the 16 accumulate
// FMAs per l-pair are per polarisation pair (28 -> 22 instructions per ring pair, l-pair and polarisation pair)
template <int R>
__global__ void __launch_bounds__(256) k_leg2_synth_np2(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                        const double* __restrict__ st, int npol, int ip0,
                                                        double* __restrict__ ph, int64_t ph_stride, int kq) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m), chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw), lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lane = threadIdx.x & 63;
    const int lmax = A.lmax;
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    const double* __restrict__ as = st + 4 * ((int64_t)npol * (mo - m) + ip0);
    const int64_t ls4 = 4 * (int64_t)npol;
    Leg2State<R> S;
    leg2_load_state<R>(A, m, chunk, lane, S);
    double ar[R][2][4], ai[R][2][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) ar[r][p][k] = ai[r][p][k] = 0.0;
        if (S.ls[r] == lw) { S.pc[r] = S.sd(r, 0); S.pp[r] = S.sd(r, 1); S.mc[r] = S.sd(r, 2); S.mp[r] = S.sd(r, 3); }
    }
    for (int l = lw; l <= lmax; l += 2) {
        const double* __restrict__ c0 = as + ls4 * l;
        const double* __restrict__ c1 = c0 + ls4;
        const double al1 = al[l + 1], be1 = be[l + 1], al2 = al[l + 2], be2 = be[l + 2];
        const bool inj = l < lAend;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double W = S.pc[r] + S.mc[r], X = S.pc[r] - S.mc[r];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const double e0r = c0[4 * p], e0i = c0[4 * p + 1], b0r = c0[4 * p + 2], b0i = c0[4 * p + 3];
                ar[r][p][0] += e0r * W;  ai[r][p][0] += e0i * W;
                ar[r][p][1] -= b0i * X;  ai[r][p][1] += b0r * X;
                ar[r][p][2] += b0r * W;  ai[r][p][2] += b0i * W;
                ar[r][p][3] += e0i * X;  ai[r][p][3] -= e0r * X;
            }
            if (inj) leg2_advance<R, true>(S, r, l, al1, be1); else leg2_advance<R, false>(S, r, l, al1, be1);
            W = S.pc[r] + S.mc[r];
            X = S.pc[r] - S.mc[r];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const double e1r = c1[4 * p], e1i = c1[4 * p + 1], b1r = c1[4 * p + 2], b1i = c1[4 * p + 3];
                ar[r][p][1] += e1r * W;  ai[r][p][1] += e1i * W;
                ar[r][p][0] -= b1i * X;  ai[r][p][0] += b1r * X;
                ar[r][p][3] += b1r * W;  ai[r][p][3] += b1i * W;
                ar[r][p][2] += e1i * X;  ai[r][p][2] -= e1r * X;
            }
            if (inj) leg2_advance<R, true>(S, r, l + 1, al2, be2); else leg2_advance<R, false>(S, r, l + 1, al2, be2);
        }
    }
    const int l0 = m > 2 ? m : 2;
    const bool swap = ((l0 + m) & 1) != 0;
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int pr = base + r * 64;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double kr = swap ? ar[r][p][2 * q + 1] : ar[r][p][2 * q], ki = swap ? ai[r][p][2 * q + 1] : ai[r][p][2 * q];
                const double fr = swap ? ar[r][p][2 * q] : ar[r][p][2 * q + 1], fi = swap ? ai[r][p][2 * q] : ai[r][p][2 * q + 1];
                double* o = ph + (kq + 2 * p + q) * ph_stride + d_phidx(A.lmax + 1, pr, m);
                o[0] = kr + fr;
                o[1] = ki + fi;
                o[2] = kr - fr;
                o[3] = ki - fi;
            }
    }
}

// ---- NP polarisation pairs per wave at R ring pairs per lane (round 3; the np2 kernel above is <2, 2>): with three or
// four pairs sharing the two recursions and W, X the work per pair and step falls from 11 to 10 / 9.5 VALU operations;
// one ring pair per lane keeps the 16 NP R accumulators within the register budget of three waves per SIMD.
template <int R, int NP>
__global__ void __launch_bounds__(256) k_leg2_synth_npx(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                        const double* __restrict__ st, int npol, int ip0,
                                                        double* __restrict__ ph, int64_t ph_stride, int kq, int split) {
    // split = A.R / R: a task of the plan's list (64 A.R ring pairs) is shared by `split` waves of 64 R pairs each
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = (blockIdx.x / split) * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk) * split + (int)(blockIdx.x % split);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw), lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lane = threadIdx.x & 63;
    const int lmax = A.lmax;
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    const double* __restrict__ as = st + 4 * ((int64_t)npol * (mo - m) + ip0);
    const int64_t ls4 = 4 * (int64_t)npol;
    Leg2State<R> S;
    leg2_load_state<R>(A, m, chunk, lane, S);
    double ar[R][NP][4], ai[R][NP][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) ar[r][p][k] = ai[r][p][k] = 0.0;
        if (S.ls[r] == lw) { S.pc[r] = S.sd(r, 0); S.pp[r] = S.sd(r, 1); S.mc[r] = S.sd(r, 2); S.mp[r] = S.sd(r, 3); }
    }
    for (int l = lw; l <= lmax; l += 2) {
        const double* __restrict__ c0 = as + ls4 * l;
        const double* __restrict__ c1 = c0 + ls4;
        const double al1 = al[l + 1], be1 = be[l + 1], al2 = al[l + 2], be2 = be[l + 2];
        const bool inj = l < lAend;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double W = S.pc[r] + S.mc[r], X = S.pc[r] - S.mc[r];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const double e0r = c0[4 * p], e0i = c0[4 * p + 1], b0r = c0[4 * p + 2], b0i = c0[4 * p + 3];
                ar[r][p][0] += e0r * W;  ai[r][p][0] += e0i * W;
                ar[r][p][1] -= b0i * X;  ai[r][p][1] += b0r * X;
                ar[r][p][2] += b0r * W;  ai[r][p][2] += b0i * W;
                ar[r][p][3] += e0i * X;  ai[r][p][3] -= e0r * X;
            }
            if (inj) leg2_advance<R, true>(S, r, l, al1, be1); else leg2_advance<R, false>(S, r, l, al1, be1);
            W = S.pc[r] + S.mc[r];
            X = S.pc[r] - S.mc[r];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const double e1r = c1[4 * p], e1i = c1[4 * p + 1], b1r = c1[4 * p + 2], b1i = c1[4 * p + 3];
                ar[r][p][1] += e1r * W;  ai[r][p][1] += e1i * W;
                ar[r][p][0] -= b1i * X;  ai[r][p][0] += b1r * X;
                ar[r][p][3] += b1r * W;  ai[r][p][3] += b1i * W;
                ar[r][p][2] += e1i * X;  ai[r][p][2] -= e1r * X;
            }
            if (inj) leg2_advance<R, true>(S, r, l + 1, al2, be2); else leg2_advance<R, false>(S, r, l + 1, al2, be2);
        }
    }
    const int l0 = m > 2 ? m : 2;
    const bool swap = ((l0 + m) & 1) != 0;
    const int base = chunk * 64 * R + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int pr = base + r * 64;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double kr = swap ? ar[r][p][2 * q + 1] : ar[r][p][2 * q], ki = swap ? ai[r][p][2 * q + 1] : ai[r][p][2 * q];
                const double fr = swap ? ar[r][p][2 * q] : ar[r][p][2 * q + 1], fi = swap ? ai[r][p][2 * q] : ai[r][p][2 * q + 1];
                double* o = ph + (kq + 2 * p + q) * ph_stride + d_phidx(A.lmax + 1, pr, m);
                o[0] = kr + fr;
                o[1] = ki + fi;
                o[2] = kr - fr;
                o[3] = ki - fi;
            }
    }
}

template <int R>
__global__ void __launch_bounds__(256) k_leg2_adj_np2(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                      const double* __restrict__ ph, int64_t ph_stride, int kq,
                                                      double* __restrict__ part, int64_t part_pol_stride,
                                                      int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double lds[4][2 * kRedTile];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wid;
    if (t >= ntasks) return;
    const WaveTask T = tasks[t];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m), chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw), lAend = __builtin_amdgcn_readfirstlane(T.lAend);
    const int lmax = A.lmax;
    Leg2State<R> S;
    Adj2G<R> G[2];
    leg2_load_state<R>(A, m, chunk, lane, S);
    leg2_adj_load<R>(A, ph, ph_stride, kq, m, chunk, lane, G[0]);
    leg2_adj_load<R>(A, ph, ph_stride, kq + 2, m, chunk, lane, G[1]);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (S.ls[r] == lw) { S.pc[r] = S.sd(r, 0); S.pp[r] = S.sd(r, 1); S.mc[r] = S.sd(r, 2); S.mp[r] = S.sd(r, 3); }
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    double* __restrict__ out = part + chunk * part_chunk_stride + 4 * (mo - m);
    double* wl = lds[wid];
    const int id = lane >> 2;
    int buf = 0;
    double a4[4], b4[4];                   // (alpha, beta)_{l0g + 1 + j} of the current group (sload_d4: see k_leg_adj)
    {
        sgpr8 a = sload_d4(al + lw + 1), b = sload_d4(be + lw + 1);
        double dummy = 0.0;
        sload_wait(a, b, dummy);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a4[j] = sgpr_double(a, j); b4[j] = sgpr_double(b, j); }
    }
    for (int l0g = lw; l0g <= lmax; l0g += 4) {
        sgpr8 an = sload_d4(al + l0g + 5), bn = sload_d4(be + l0g + 5);
        double v[2][16];
        const bool inj = l0g < lAend;
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const int l = l0g + j;
            double a0[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, a1[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double W = S.pc[r] + S.mc[r], X = S.pc[r] - S.mc[r];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    a0[p][0] += W * G[p].qk_r[r] - X * G[p].uf_i[r];
                    a0[p][1] += W * G[p].qk_i[r] + X * G[p].uf_r[r];
                    a0[p][2] += W * G[p].uk_r[r] + X * G[p].qf_i[r];
                    a0[p][3] += W * G[p].uk_i[r] - X * G[p].qf_r[r];
                }
                if (inj) leg2_advance<R, true>(S, r, l, a4[j], b4[j]); else leg2_advance<R, false>(S, r, l, a4[j], b4[j]);
                W = S.pc[r] + S.mc[r];
                X = S.pc[r] - S.mc[r];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    a1[p][0] += W * G[p].qf_r[r] - X * G[p].uk_i[r];
                    a1[p][1] += W * G[p].qf_i[r] + X * G[p].uk_r[r];
                    a1[p][2] += W * G[p].uf_r[r] + X * G[p].qk_i[r];
                    a1[p][3] += W * G[p].uf_i[r] - X * G[p].qk_r[r];
                }
                if (inj) leg2_advance<R, true>(S, r, l + 1, a4[j + 1], b4[j + 1]); else leg2_advance<R, false>(S, r, l + 1, a4[j + 1], b4[j + 1]);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[p][4 * j + k] = a0[p][k]; v[p][4 * j + 4 + k] = a1[p][k]; }
        }
        const int l = l0g + (id >> 2);
        double last = 0.0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const double sacc = wave_reduce16(v[p], wl + buf * kRedTile, lane);
            buf ^= 1;
            if ((lane & 3) == 0 && l <= lmax) out[p * part_pol_stride + 4 * l + (id & 3)] = sacc;
            last = sacc;
        }
        sload_wait(an, bn, last);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a4[j] = sgpr_double(an, j); b4[j] = sgpr_double(bn, j); }
    }
}

// ---- spin-2 adjoint on the matrix unit (round 3) -----------------------------------------------------------------
// (E', B') += sum_pairs [ mu+_l (Bw + Bx) + mu-_l (Bw - Bx) ] in the mu+- basis (no W = mu+ + mu-, X = mu+ - mu- adds):
//     l + m even : Bw = (Q+.re, Q+.im, U+.re, U+.im),  Bx = (-U-.im, U-.re, Q-.im, -Q-.re)      Q+- = Q_N +- Q_S
//     l + m odd  : Bw = (Q-.re, Q-.im, U-.re, U-.im),  Bx = (-U+.im, U+.re, Q+.im, -Q+.re)
// is the contraction of k_leg_adj_mx with TWO A operands (the mu+ and the mu- tile) and 4 output columns per
// polarisation pair: with four pairs in a launch (polarised multi-band runs) the 16 MFMA columns are full, the two
// spin-weighted recursions are shared by the four pairs and the wave-wide reduction of k_leg2_adj (wave_reduce16 per
// 4 l: ~40 % of its instructions) is gone.  Same task shape as k_leg_adj_mx (2 waves = one (m, 128 NR pairs) task,
// wave-private LDS tile [32 l][64 pairs], B operands in registers for the whole task); the two chains go through the
// tile one after the other, so the LDS footprint -- which sets the occupancy -- stays that of the scalar kernel.
//   lane i: A[row = i & 15][k = i >> 4],  B[k = i >> 4][col = i & 15],  col = 4 (pair of the launch) + component.
template <bool INJECT, bool MINUS>
__device__ __forceinline__ void mx2_recur(const double* __restrict__ al, const double* __restrict__ be, int l0, double x,
                                          double& cur, double& prev, int ls, const double* __restrict__ sd,
                                          double* __restrict__ trow) {
#pragma unroll
    for (int j = 0; j < kMxL; ++j) {
        const double a1 = al[l0 + j + 1], b1 = be[l0 + j + 1];
        if (INJECT) if (ls == l0 + j) { cur = sd[MINUS ? 2 : 0]; prev = sd[MINUS ? 3 : 1]; }
        trow[j * kMxPitch] = cur;
        const double t = MINUS ? a1 * x - b1 : a1 * x + b1;
        const double n = t * cur - prev;
        prev = cur;
        cur = n;
    }
}

template <int NR>
__global__ void __launch_bounds__(128) k_leg2_adj_mx(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                     const double* __restrict__ ph, int64_t ph_stride, int kq0, int nb,
                                                     double* __restrict__ part, int64_t part_pol_stride,
                                                     int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double tile[2][kMxL * kMxPitch];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= ntasks) return;
    const WaveTask T = tasks[blockIdx.x];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lmax = A.lmax;
    int pb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sub = NR == 1 ? wid : (wid == 0 ? 3 * r : 1 + r);
        pb[r] = chunk * (128 * NR) + sub * 64;
    }
    double x[NR], pc[NR], pp[NR], mc[NR], mp[NR];
    const double* sd[NR];
    int ls[NR], lwr[NR], lhi[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = pb[r] + lane;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x[r] = A.x[p];
        ls[r] = A.ls[idx];
        sd[r] = A.seed + idx * 4;
        pc[r] = pp[r] = mc[r] = mp[r] = 0.0;
        int v = ls[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        lwr[r] = __builtin_amdgcn_readfirstlane(v);
        v = ls[r] == 0x3fffffff ? -1 : ls[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        lhi[r] = __builtin_amdgcn_readfirstlane(v);
    }
    // B operands.  Lane (kq, col): polarisation pair ipp = col >> 2, component c = col & 3 (E'r, E'i, B'r, B'i).
    //   Bw source a = (c < 2 ? Q : U) part (c & 1);  Bx source b = sB * (c < 2 ? U : Q) part 1 - (c & 1), sB = -1 for c = 0, 3
    const int kq = lane >> 4, col = lane & 15, ipp = col >> 2, c = col & 3;
    const bool on = ipp < nb;
    const int par0 = (lw + m) & 1;                     // parity class (l + m) & 1 of the even rows (l - l0 even)
    double Bpe[NR][16], Bme[NR][16], Bpo[NR][16], Bmo[NR][16];   // mu+ / mu- operands of the even / odd rows
    {
        const int mapA = kq0 + 2 * (on ? ipp : 0) + (c < 2 ? 0 : 1), mapB = kq0 + 2 * (on ? ipp : 0) + (c < 2 ? 1 : 0);
        const double* __restrict__ ga = ph + (int64_t)mapA * ph_stride + (c & 1);
        const double* __restrict__ gb = ph + (int64_t)mapB * ph_stride + (1 - (c & 1));
        const double sB = (c == 0 || c == 3) ? -1.0 : 1.0;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t o = d_phidx(lmax + 1, pb[r] + kq + 4 * q, m);
                const double an = on ? ga[o] : 0.0, as = on ? ga[o + 2] : 0.0;
                const double bn = on ? gb[o] : 0.0, bs = on ? gb[o + 2] : 0.0;
                const double ap = an + as, am = an - as, bp = sB * (bn + bs), bm = sB * (bn - bs);
                // class 0 (l + m even): Bw = ap, Bx = bm;  class 1: Bw = am, Bx = bp
                const double w0 = ap, x0 = bm, w1 = am, x1 = bp;
                const double we = par0 ? w1 : w0, xe = par0 ? x1 : x0, wo = par0 ? w0 : w1, xo = par0 ? x0 : x1;
                Bpe[r][q] = we + xe; Bme[r][q] = we - xe;
                Bpo[r][q] = wo + xo; Bmo[r][q] = wo - xo;
            }
    }
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    double* __restrict__ outp = part + (int64_t)(on ? ipp : 0) * part_pol_stride + chunk * part_chunk_stride + 4 * (mo - m) + c;
    double* __restrict__ Tw = tile[wid];
    const double* __restrict__ To = tile[1 - wid];
    const int arow = (lane & 15) * 2 * kMxPitch + kq;
    for (int l0 = lw; l0 <= lmax; l0 += kMxL) {
        mx_d4 De0 = {0.0, 0.0, 0.0, 0.0}, De1 = De0, Do0 = De0, Do1 = De0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (l0 + kMxL <= lwr[r]) continue;            // no pair of this sub-block has started yet (wave-uniform)
#define CMDR_MX2_Q(q, BE, BO)                                                                              \
            {                                                                                              \
                const double ae0 = Tw[arow + 4 * (q)], ao0 = Tw[arow + kMxPitch + 4 * (q)];                \
                const double ae1 = Tw[arow + 4 * (q) + 4], ao1 = Tw[arow + kMxPitch + 4 * (q) + 4];        \
                De0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ae0, BE[r][(q)], De0, 0, 0, 0);                 \
                Do0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ao0, BO[r][(q)], Do0, 0, 0, 0);                 \
                De1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ae1, BE[r][(q) + 1], De1, 0, 0, 0);             \
                Do1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ao1, BO[r][(q) + 1], Do1, 0, 0, 0);             \
            }
#define CMDR_MX2_ALL(BE, BO) CMDR_MX2_Q(0, BE, BO) CMDR_MX2_Q(2, BE, BO) CMDR_MX2_Q(4, BE, BO) CMDR_MX2_Q(6, BE, BO) \
                             CMDR_MX2_Q(8, BE, BO) CMDR_MX2_Q(10, BE, BO) CMDR_MX2_Q(12, BE, BO) CMDR_MX2_Q(14, BE, BO)
#define CMDR_WAVE_SYNC()                                                                                   \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                         \
            __builtin_amdgcn_wave_barrier();                                                               \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool inj = l0 <= lhi[r];                // seeds only in the groups in which a lane of this block starts
            // ---- mu+ through the tile
            if (inj) mx2_recur<true, false>(al, be, l0, x[r], pc[r], pp[r], ls[r], sd[r], Tw + lane);
            else     mx2_recur<false, false>(al, be, l0, x[r], pc[r], pp[r], ls[r], sd[r], Tw + lane);
            CMDR_WAVE_SYNC()
            CMDR_MX2_ALL(Bpe, Bpo)
            CMDR_WAVE_SYNC()
            // ---- mu- through the same tile
            if (inj) mx2_recur<true, true>(al, be, l0, x[r], mc[r], mp[r], ls[r], sd[r], Tw + lane);
            else     mx2_recur<false, true>(al, be, l0, x[r], mc[r], mp[r], ls[r], sd[r], Tw + lane);
            CMDR_WAVE_SYNC()
            CMDR_MX2_ALL(Bme, Bmo)
            CMDR_WAVE_SYNC()
#undef CMDR_MX2_Q
#undef CMDR_MX2_ALL
#undef CMDR_WAVE_SYNC
        }
        De0 += De1;
        Do0 += Do1;
        // wave 0 finishes the even rows (l - l0 even), wave 1 the odd ones: hand the other half over
        const mx_d4 give = wid == 0 ? Do0 : De0;
        mx_d4 keep = wid == 0 ? De0 : Do0;
#pragma unroll
        for (int v = 0; v < 4; ++v) Tw[v * 64 + lane] = give[v];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v) keep[v] += To[v * 64 + lane];
        if (on) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int l = l0 + 2 * (kq + 4 * v) + wid;
                if (l <= lmax) outp[4 * l] = keep[v];
            }
        }
        __syncthreads();
    }
}

// ---- spin-2 adjoint of one or two (Q,U) pairs without the matrix unit: the task, recursions and LDS transposition of
// k_leg2_adj_mx, the accumulation of k_leg_adj_dx: lane = (l row, pair kq of quad q); per pair of the launch and 64-pair
// block the 16 B values (mu+ / mu- operand x even / odd rows x 4 components) of ring pair 4 row + kq sit in registers and
// step q takes lane q of every 16-lane row inside the FMA (v_fmac_f64_dpp row_newbcast:q).  8 + 16 NP flop per (ring
// pair, l) -- the VALU kernels' count without W / X and without the wave-wide reduction (wave_reduce16 per 4 l).
template <int NR, int NP>
__global__ void __launch_bounds__(128) k_leg2_adj_dx(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                     const double* __restrict__ ph, int64_t ph_stride, int kq0,
                                                     double* __restrict__ part, int64_t part_pol_stride,
                                                     int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double tile[2][kMxL * kMxPitch];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= ntasks) return;
    const WaveTask T = tasks[blockIdx.x];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lmax = A.lmax;
    const int kq = lane >> 4, row = lane & 15;
    int pb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int sub = NR == 1 ? wid : (wid == 0 ? 3 * r : 1 + r);
        pb[r] = chunk * (128 * NR) + sub * 64;
    }
    double x[NR], pc[NR], pp[NR], mc[NR], mp[NR];
    const double* sd[NR];
    int ls[NR], lwr[NR], lhi[NR];
    const int par0 = (lw + m) & 1;
    // B[r][p][s][e/o][c]: s = 0 the mu+ operand (Bw + Bx), 1 the mu- operand (Bw - Bx); see k_leg2_adj_mx
    double B[NR][NP][2][2][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int p = pb[r] + lane;
        const int64_t idx = (int64_t)m * A.npair_pad + p;
        x[r] = A.x[p];
        ls[r] = A.ls[idx];
        sd[r] = A.seed + idx * 4;
        pc[r] = pp[r] = mc[r] = mp[r] = 0.0;
        int v = ls[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
        lwr[r] = __builtin_amdgcn_readfirstlane(v);
        v = ls[r] == 0x3fffffff ? -1 : ls[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
        lhi[r] = __builtin_amdgcn_readfirstlane(v);
        const int64_t o = d_phidx(lmax + 1, pb[r] + 4 * row + kq, m);
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) {
            const double* q = ph + (int64_t)(kq0 + 2 * ip) * ph_stride + o;
            const double* u = q + ph_stride;
            const double qpr = q[0] + q[2], qpi = q[1] + q[3], qmr = q[0] - q[2], qmi = q[1] - q[3];
            const double upr = u[0] + u[2], upi = u[1] + u[3], umr = u[0] - u[2], umi = u[1] - u[3];
            // class 0 (l + m even): Bw = (Q+r, Q+i, U+r, U+i), Bx = (-U-i, U-r, Q-i, -Q-r); class 1: + and - exchanged
            const double w0[4] = {qpr, qpi, upr, upi}, x0[4] = {-umi, umr, qmi, -qmr};
            const double w1[4] = {qmr, qmi, umr, umi}, x1[4] = {-upi, upr, qpi, -qpr};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double we = par0 ? w1[c] : w0[c], xe = par0 ? x1[c] : x0[c];
                const double wo = par0 ? w0[c] : w1[c], xo = par0 ? x0[c] : x1[c];
                B[r][ip][0][0][c] = we + xe; B[r][ip][1][0][c] = we - xe;
                B[r][ip][0][1][c] = wo + xo; B[r][ip][1][1][c] = wo - xo;
            }
        }
    }
    const int64_t mo = d_moffp(lmax, m);
    const double* __restrict__ al = A.alpha + (mo - m);
    const double* __restrict__ be = A.beta + (mo - m);
    // after the fold the 16-lane row kq of a lane holds component {0, 2, 1, 3}[kq] (swap_halves / swap_rows below)
    const int cidx = ((kq & 1) << 1) | (kq >> 1);
    double* __restrict__ outp = part + chunk * part_chunk_stride + 4 * (mo - m) + cidx;
    double* __restrict__ Tw = tile[wid];
    const double* __restrict__ To = tile[1 - wid];
    const int arow = row * 2 * kMxPitch + kq;
    for (int l0 = lw; l0 <= lmax; l0 += kMxL) {
        double ae_[NP][4], ao_[NP][4];
#pragma unroll
        for (int ip = 0; ip < NP; ++ip)
#pragma unroll
            for (int c = 0; c < 4; ++c) ae_[ip][c] = ao_[ip][c] = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (l0 + kMxL <= lwr[r]) continue;
            const bool inj = l0 <= lhi[r];
#define CMDR_DX2_Q(q, S)                                                                                   \
            {                                                                                              \
                const double ae = Tw[arow + 4 * (q)], ao = Tw[arow + kMxPitch + 4 * (q)];                  \
                _Pragma("unroll") for (int ip = 0; ip < NP; ++ip)                                          \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                            \
                    fmac_row_bcast<(q)>(ae_[ip][c], B[r][ip][S][0][c], ae);                                \
                    fmac_row_bcast<(q)>(ao_[ip][c], B[r][ip][S][1][c], ao);                                \
                }                                                                                          \
            }
#define CMDR_DX2_ALL(S) CMDR_DX2_Q(0, S) CMDR_DX2_Q(1, S) CMDR_DX2_Q(2, S) CMDR_DX2_Q(3, S) CMDR_DX2_Q(4, S) CMDR_DX2_Q(5, S) \
                        CMDR_DX2_Q(6, S) CMDR_DX2_Q(7, S) CMDR_DX2_Q(8, S) CMDR_DX2_Q(9, S) CMDR_DX2_Q(10, S) CMDR_DX2_Q(11, S) \
                        CMDR_DX2_Q(12, S) CMDR_DX2_Q(13, S) CMDR_DX2_Q(14, S) CMDR_DX2_Q(15, S)
#define CMDR_WAVE_SYNC()                                                                                   \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                         \
            __builtin_amdgcn_wave_barrier();                                                               \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (inj) mx2_recur<true, false>(al, be, l0, x[r], pc[r], pp[r], ls[r], sd[r], Tw + lane);
            else     mx2_recur<false, false>(al, be, l0, x[r], pc[r], pp[r], ls[r], sd[r], Tw + lane);
            CMDR_WAVE_SYNC()
            CMDR_DX2_ALL(0)
            CMDR_WAVE_SYNC()
            if (inj) mx2_recur<true, true>(al, be, l0, x[r], mc[r], mp[r], ls[r], sd[r], Tw + lane);
            else     mx2_recur<false, true>(al, be, l0, x[r], mc[r], mp[r], ls[r], sd[r], Tw + lane);
            CMDR_WAVE_SYNC()
            CMDR_DX2_ALL(1)
            CMDR_WAVE_SYNC()
#undef CMDR_DX2_Q
#undef CMDR_DX2_ALL
#undef CMDR_WAVE_SYNC
        }
        // fold the four pair lanes (kq) of every row: te / to hold, in 16-lane row kq, component {0, 2, 1, 3}[kq] of the
        // even / odd l rows; this wave keeps the parity it writes and hands the other one over
        double keep[NP];
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) {
            double a0 = ae_[ip][0], a1 = ae_[ip][1], a2 = ae_[ip][2], a3 = ae_[ip][3];
            swap_halves(a0, a1);
            swap_halves(a2, a3);
            double s01 = a0 + a1, s23 = a2 + a3;
            swap_rows(s01, s23);
            const double te = s01 + s23;
            a0 = ao_[ip][0]; a1 = ao_[ip][1]; a2 = ao_[ip][2]; a3 = ao_[ip][3];
            swap_halves(a0, a1);
            swap_halves(a2, a3);
            s01 = a0 + a1; s23 = a2 + a3;
            swap_rows(s01, s23);
            const double to = s01 + s23;
            keep[ip] = wid == 0 ? te : to;
            Tw[ip * 64 + lane] = wid == 0 ? to : te;
        }
        __syncthreads();
        const int l = l0 + 2 * row + wid;
        if (l <= lmax) {
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) outp[(int64_t)ip * part_pol_stride + 4 * l] = keep[ip] + To[ip * 64 + lane];
        }
        __syncthreads();
    }
}

// ---- the DPP form, software-pipelined: the recursion that fills the NEXT tile is interleaved, instruction by instruction,
// with the DPP FMAs that consume the CURRENT one (two tiles per wave, ping-pong: tile 0 always carries the + chain, tile 1
// the - chain).  k_leg2_adj_dx runs [recursion -> fence -> 16 x FMA block -> fence] twice per 32 l with nothing else in
// flight, and its recursion phase is a dependent chain that also pushes 8 B per lane and l through the LDS store path;
// here the chain's latency and the store path hide behind the accumulation of the other chain.  One wave per SIMD (66 KB
// of LDS per workgroup), so everything has to come from the wave's own instruction stream -- and nothing in the loop may
// wait for a scalar load (SMEM returns out of order: every wait on it is lgkmcnt(0), which also drains the LDS queue).
// Hence the sign-alternated recursion: with nu_l = s_k mu_l, k = l - lw, s = (+, +, -, -, ...),
//     nu_{l+1} = nu_{l-1} + (-1)^k (alpha_{l+1} x +- beta_{l+1}) nu_l,
// the new value ACCUMULATES into the register of nu_{l-1}, which is dead afterwards, so a step is one multiply (x nu_l)
// and two v_fmac_f64_dpp whose row-broadcast operand is the pre-signed coefficient (Leg2Args::abs_) held by lane l of
// every 16-lane row: one coalesced vector load per 16 l (vmcnt: counted, in order) instead of 32 scalar loads.  The tile
// carries nu; s_k = (-1)^row for both parities of a 32-l group, applied once when the column is stored.
// tile of the pipelined kernel: 32 rows x 64 ring pairs, the 16 even-(l - l0) rows first, then the 16 odd ones (the two
// values a lane reads per FMA block, and the two a lane stores per recursion stage, are then 16 rows = 8.4 KB apart: the
// compiler cannot fuse them into ds_read2_b64 / ds_write2_b64, which cost 8 / 13 LDS cycles against 2 x 2 / 2 x 6 for the
// single forms, MI355X_MICROARCH.md); pitch 66 doubles: (2 x 66) mod 64 = 4 banks per row -> the 32 lanes of a
// ds_read_b64 group (16 rows x 2 pairs) hit distinct banks
constexpr int kPxPitch = 66;
template <int Q, bool NEG>
__device__ __forceinline__ void fmac_row_bcast_n(double& acc, double g, double a) {
    if (NEG) asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(a), "n"(Q));
    else     asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(a), "n"(Q));
}
// recursion steps j = J0, J0 + 1 (J0 even) of one chain: on entry cur = nu_{l0 + J0}, prev = nu_{l0 + J0 - 1}; ca / cb hold
// (alphaS, betaS)_{l0 + 1 + 16 h + (lane & 15)} for h = J0 / 16
template <bool INJECT, bool MINUS, int J0>
__device__ __forceinline__ void px_recur2(int l0, double x, double& cur, double& prev, int ls, int lw,
                                          const double* __restrict__ sd, double ca, double cb, double* __restrict__ trow) {
    if (INJECT) {
        if (ls == l0 + J0) {          // seeds mu_ls, mu_{ls-1} -> nu: s_k with k = ls - lw, s_{k-1}
            const int k = (ls - lw) & 3;
            const double sc = (k & 2) ? -1.0 : 1.0, sp = ((k + 3) & 2) ? -1.0 : 1.0;
            cur = sc * sd[MINUS ? 2 : 0];
            prev = sp * sd[MINUS ? 3 : 1];
        }
    }
    trow[(J0 >> 1) * kPxPitch] = cur;                         // even l - l0: rows 0..15
    {
        const double xc = x * cur;
        fmac_row_bcast_n<J0 & 15, false>(prev, ca, xc);
        fmac_row_bcast_n<J0 & 15, MINUS>(prev, cb, cur);       // prev = nu_{l0 + J0 + 1}
    }
    if (INJECT) {
        if (ls == l0 + J0 + 1) {
            const int k = (ls - lw) & 3;
            const double sc = (k & 2) ? -1.0 : 1.0, sp = ((k + 3) & 2) ? -1.0 : 1.0;
            prev = sc * sd[MINUS ? 2 : 0];
            cur = sp * sd[MINUS ? 3 : 1];
        }
    }
    trow[(16 + (J0 >> 1)) * kPxPitch] = prev;                  // odd l - l0: rows 16..31
    {
        const double xc = x * prev;
        fmac_row_bcast_n<(J0 + 1) & 15, false>(cur, ca, xc);
        fmac_row_bcast_n<(J0 + 1) & 15, MINUS>(cur, cb, prev);  // cur = nu_{l0 + J0 + 2}; roles are back
    }
}

// One wave = one (m, 128 ring pairs) task: its two 64-pair blocks take turns in the pipeline -- (b0, +) (b0, -) (b1, +)
// (b1, -) per 32-l group, each phase's FMA blocks beside the recursion of the next phase -- and both accumulate into the
// same registers, so there is no second wave to wait for and nothing to hand over through LDS.
template <int NP>
__global__ void __launch_bounds__(64) k_leg2_adj_px(Leg2Args A, const WaveTask* __restrict__ tasks, int ntasks,
                                                    const double* __restrict__ ph, int64_t ph_stride, int kq0,
                                                    double* __restrict__ part, int64_t part_pol_stride,
                                                    int64_t part_chunk_stride) {
    __shared__ __attribute__((aligned(16))) double tile[2][kMxL * kPxPitch];      // ping-pong
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= ntasks) return;
    const WaveTask T = tasks[blockIdx.x];
    if (T.chunk < 0) return;
    const int m = __builtin_amdgcn_readfirstlane(T.m);
    const int chunk = __builtin_amdgcn_readfirstlane(T.chunk);
    const int lw = __builtin_amdgcn_readfirstlane(T.lw);
    const int lmax = A.lmax;
    const int kq = lane >> 4, row = lane & 15;
    const int par0 = (lw + m) & 1;
    double x0, x1, pc0 = 0.0, pp0 = 0.0, mc0 = 0.0, mp0 = 0.0, pc1 = 0.0, pp1 = 0.0, mc1 = 0.0, mp1 = 0.0;
    const double *sd0, *sd1;
    int ls0, ls1, lhi;
    double B0[NP][2][2][4], B1[NP][2][2][4];
    {
        int hi = -1;
        auto setup = [&](int b, double& x, int& ls, const double*& sd, double (&B)[NP][2][2][4]) {
            const int pb = chunk * 128 + b * 64;
            const int p = pb + lane;
            const int64_t idx = (int64_t)m * A.npair_pad + p;
            x = A.x[p];
            ls = A.ls[idx];
            sd = A.seed + idx * 4;
            int v = ls == 0x3fffffff ? -1 : ls;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
            hi = max(hi, __builtin_amdgcn_readfirstlane(v));
            const int64_t o = d_phidx(lmax + 1, pb + 4 * row + kq, m);
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) {
                const double* q = ph + (int64_t)(kq0 + 2 * ip) * ph_stride + o;
                const double* u = q + ph_stride;
                const double qpr = q[0] + q[2], qpi = q[1] + q[3], qmr = q[0] - q[2], qmi = q[1] - q[3];
                const double upr = u[0] + u[2], upi = u[1] + u[3], umr = u[0] - u[2], umi = u[1] - u[3];
                const double w0[4] = {qpr, qpi, upr, upi}, xx0[4] = {-umi, umr, qmi, -qmr};
                const double w1[4] = {qmr, qmi, umr, umi}, xx1[4] = {-upi, upr, qpi, -qpr};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double we = par0 ? w1[c] : w0[c], xe = par0 ? xx1[c] : xx0[c];
                    const double wo = par0 ? w0[c] : w1[c], xo = par0 ? xx0[c] : xx1[c];
                    B[ip][0][0][c] = we + xe; B[ip][1][0][c] = we - xe;
                    B[ip][0][1][c] = wo + xo; B[ip][1][1][c] = wo - xo;
                }
            }
        };
        setup(0, x0, ls0, sd0, B0);
        setup(1, x1, ls1, sd1, B1);
        lhi = hi;
    }
    const int64_t mo = d_moffp(lmax, m);
    // (alphaS, betaS)_{l}: this lane's entry of a 16-l block that starts at l = lb + 1 is ab[2 lb]
    const double* __restrict__ ab = A.abs_ + 2 * (mo - m) + 2 * (1 + row);
    const int cidx = ((kq & 1) << 1) | (kq >> 1);
    double* __restrict__ outp = part + chunk * part_chunk_stride + 4 * (mo - m) + cidx;
    const double osgn = (row & 1) ? -1.0 : 1.0;          // nu -> mu for the rows this lane stores (both parities)
    double* __restrict__ T0 = tile[0];
    double* __restrict__ T1 = tile[1];
    const int arow = row * kPxPitch + kq;
#define CMDR_WAVE_SYNC()                                                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                 \
    __builtin_amdgcn_wave_barrier();                                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the tile reads of FMA block q + 3 are issued before block q is consumed (LDS is in order and the compiler keeps the
    // program order of the reads against the recursion's stores, which it cannot prove disjoint)
#define CMDR_PX_RD(q, TT, E, O) { E = TT[arow + 4 * (q)]; O = TT[arow + 16 * kPxPitch + 4 * (q)]; }
#define CMDR_PX_FMA(q, BB, S, E, O)                                                                        \
    {                                                                                                      \
        _Pragma("unroll") for (int ip = 0; ip < NP; ++ip)                                                  \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                    \
            fmac_row_bcast<(q)>(ae_[ip][c], BB[ip][S][0][c], E);                                           \
            fmac_row_bcast<(q)>(ao_[ip][c], BB[ip][S][1][c], O);                                           \
        }                                                                                                  \
    }
    // one pipeline stage: FMA block q (operands E, O read three stages ago; B set BB, sign S) on tile TT beside recursion
    // steps 2q, 2q + 1 of the next phase (block state XN / CUR / PREV / LSN / SDN, sign MN, group start LN) into TN
#define CMDR_PX_STAGE(q, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, E, O, EN, ON)         \
    if ((q) + 3 < 16) CMDR_PX_RD(((q) + 3) & 15, TT, EN, ON)                                               \
    CMDR_PX_FMA(q, BB, S, E, O)                                                                            \
    px_recur2<INJ, MN, 2 * (q)>(LN, XN, CUR, PREV, LSN, lw, SDN, CA[(q) >> 3], CB[(q) >> 3], TN + lane);
#define CMDR_PX_PHASE(BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB)                          \
    {                                                                                                      \
        double e0, o0, e1, o1, e2, o2, e3, o3;                                                             \
        CMDR_PX_RD(0, TT, e0, o0) CMDR_PX_RD(1, TT, e1, o1) CMDR_PX_RD(2, TT, e2, o2)                      \
        CMDR_PX_STAGE(0, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e0, o0, e3, o3)       \
        CMDR_PX_STAGE(1, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e1, o1, e0, o0)       \
        CMDR_PX_STAGE(2, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e2, o2, e1, o1)       \
        CMDR_PX_STAGE(3, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e3, o3, e2, o2)       \
        CMDR_PX_STAGE(4, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e0, o0, e3, o3)       \
        CMDR_PX_STAGE(5, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e1, o1, e0, o0)       \
        CMDR_PX_STAGE(6, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e2, o2, e1, o1)       \
        CMDR_PX_STAGE(7, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e3, o3, e2, o2)       \
        CMDR_PX_STAGE(8, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e0, o0, e3, o3)       \
        CMDR_PX_STAGE(9, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e1, o1, e0, o0)       \
        CMDR_PX_STAGE(10, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e2, o2, e1, o1)      \
        CMDR_PX_STAGE(11, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e3, o3, e2, o2)      \
        CMDR_PX_STAGE(12, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e0, o0, e3, o3)      \
        CMDR_PX_STAGE(13, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e1, o1, e0, o0)      \
        CMDR_PX_STAGE(14, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e2, o2, e1, o1)      \
        CMDR_PX_STAGE(15, BB, S, TT, INJ, MN, LN, XN, CUR, PREV, LSN, SDN, TN, CA, CB, e3, o3, e2, o2)      \
    }
    auto load_ab = [&](int lb, double (&Ca)[2], double (&Cb)[2]) {
        // rows beyond the table's end (lb + 32 may pass lmax + 1 by up to 32 + 31) read the 64 entries of slack
        const double2 v0 = *reinterpret_cast<const double2*>(ab + 2 * lb);
        const double2 v1 = *reinterpret_cast<const double2*>(ab + 2 * (lb + 16));
        Ca[0] = v0.x; Cb[0] = v0.y;
        Ca[1] = v1.x; Cb[1] = v1.y;
    };
    double Ca[2], Cb[2], Na[2], Nb[2];
    // prologue: the + chain of block 0 for the first group, un-pipelined
    load_ab(lw, Ca, Cb);
    {
#define CMDR_PX_PRO(q, INJ) px_recur2<INJ, false, 2 * (q)>(lw, x0, pc0, pp0, ls0, lw, sd0, Ca[(q) >> 3], Cb[(q) >> 3], T0 + lane);
        if (lw <= lhi) { CMDR_PX_PRO(0, true) CMDR_PX_PRO(1, true) CMDR_PX_PRO(2, true) CMDR_PX_PRO(3, true) CMDR_PX_PRO(4, true)
                         CMDR_PX_PRO(5, true) CMDR_PX_PRO(6, true) CMDR_PX_PRO(7, true) CMDR_PX_PRO(8, true) CMDR_PX_PRO(9, true)
                         CMDR_PX_PRO(10, true) CMDR_PX_PRO(11, true) CMDR_PX_PRO(12, true) CMDR_PX_PRO(13, true)
                         CMDR_PX_PRO(14, true) CMDR_PX_PRO(15, true) }
        else           { CMDR_PX_PRO(0, false) CMDR_PX_PRO(1, false) CMDR_PX_PRO(2, false) CMDR_PX_PRO(3, false) CMDR_PX_PRO(4, false)
                         CMDR_PX_PRO(5, false) CMDR_PX_PRO(6, false) CMDR_PX_PRO(7, false) CMDR_PX_PRO(8, false) CMDR_PX_PRO(9, false)
                         CMDR_PX_PRO(10, false) CMDR_PX_PRO(11, false) CMDR_PX_PRO(12, false) CMDR_PX_PRO(13, false)
                         CMDR_PX_PRO(14, false) CMDR_PX_PRO(15, false) }
#undef CMDR_PX_PRO
        CMDR_WAVE_SYNC()
    }
    for (int l0 = lw; l0 <= lmax; l0 += kMxL) {
        double ae_[NP][4], ao_[NP][4];
#pragma unroll
        for (int ip = 0; ip < NP; ++ip)
#pragma unroll
            for (int c = 0; c < 4; ++c) ae_[ip][c] = ao_[ip][c] = 0.0;
        const bool inj = l0 <= lhi, injn = l0 + kMxL <= lhi;
        load_ab(l0 + kMxL, Na, Nb);                    // the next group's coefficients: in flight during this group
        // (b0, +) on tile 0 | the - chain of block 0 -> tile 1
        if (inj) { CMDR_PX_PHASE(B0, 0, T0, true, true, l0, x0, mc0, mp0, ls0, sd0, T1, Ca, Cb) }
        else     { CMDR_PX_PHASE(B0, 0, T0, false, true, l0, x0, mc0, mp0, ls0, sd0, T1, Ca, Cb) }
        CMDR_WAVE_SYNC()
        // (b0, -) on tile 1 | the + chain of block 1 -> tile 0
        if (inj) { CMDR_PX_PHASE(B0, 1, T1, true, false, l0, x1, pc1, pp1, ls1, sd1, T0, Ca, Cb) }
        else     { CMDR_PX_PHASE(B0, 1, T1, false, false, l0, x1, pc1, pp1, ls1, sd1, T0, Ca, Cb) }
        CMDR_WAVE_SYNC()
        // (b1, +) on tile 0 | the - chain of block 1 -> tile 1
        if (inj) { CMDR_PX_PHASE(B1, 0, T0, true, true, l0, x1, mc1, mp1, ls1, sd1, T1, Ca, Cb) }
        else     { CMDR_PX_PHASE(B1, 0, T0, false, true, l0, x1, mc1, mp1, ls1, sd1, T1, Ca, Cb) }
        CMDR_WAVE_SYNC()
        // (b1, -) on tile 1 | the + chain of block 0 for the NEXT group -> tile 0 (harmless beyond lmax: the tables carry
        // slack and nothing of that group is stored)
        if (injn) { CMDR_PX_PHASE(B1, 1, T1, true, false, l0 + kMxL, x0, pc0, pp0, ls0, sd0, T0, Na, Nb) }
        else      { CMDR_PX_PHASE(B1, 1, T1, false, false, l0 + kMxL, x0, pc0, pp0, ls0, sd0, T0, Na, Nb) }
        CMDR_WAVE_SYNC()
        Ca[0] = Na[0]; Ca[1] = Na[1]; Cb[0] = Nb[0]; Cb[1] = Nb[1];
        // fold the four pair lanes (kq) of every row; te / to hold, in 16-lane row kq, component {0, 2, 1, 3}[kq] of the
        // even / odd l rows
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) {
            double a0 = ae_[ip][0], a1 = ae_[ip][1], a2 = ae_[ip][2], a3 = ae_[ip][3];
            swap_halves(a0, a1);
            swap_halves(a2, a3);
            double s01 = a0 + a1, s23 = a2 + a3;
            swap_rows(s01, s23);
            const double te = s01 + s23;
            a0 = ao_[ip][0]; a1 = ao_[ip][1]; a2 = ao_[ip][2]; a3 = ao_[ip][3];
            swap_halves(a0, a1);
            swap_halves(a2, a3);
            s01 = a0 + a1; s23 = a2 + a3;
            swap_rows(s01, s23);
            const double to = s01 + s23;
            const int l = l0 + 2 * row;
            if (l <= lmax) outp[(int64_t)ip * part_pol_stride + 4 * l] = osgn * te;
            if (l + 1 <= lmax) outp[(int64_t)ip * part_pol_stride + 4 * (l + 1)] = osgn * to;
        }
    }
#undef CMDR_PX_PHASE
#undef CMDR_PX_STAGE
#undef CMDR_PX_FMA
#undef CMDR_PX_RD
#undef CMDR_WAVE_SYNC
}

// CMDR_LEG2_NP_S=1 / CMDR_LEG2_NP_A=1 switch the two-pairs-per-wave synthesis / adjoint off
static bool leg2_pairs2(bool adjoint) {
    static int v[2] = {-1, -1};
    if (v[adjoint] < 0) {
        v[adjoint] = 1;
        if (const char* e = std::getenv(adjoint ? "CMDR_LEG2_NP_A" : "CMDR_LEG2_NP_S")) v[adjoint] = std::atoi(e) >= 2 ? 1 : 0;
    }
    return v[adjoint] == 1;
}

void launch_leg2_synth(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* st, int npol, double* ph,
                       int64_t ph_stride, int kq0, hipStream_t s) {
    if (ntasks == 0) return;
    // three and more pairs: four (three) per wave at one ring pair per lane (k_leg2_synth_npx<1, NP>; CMDR_SYNTH2_NP=0: off)
    const bool npx_on = [] { const char* e = std::getenv("CMDR_SYNTH2_NP"); return !e || std::atoi(e) != 0; }();   // per call (test hook)
    int ip0 = 0;
    if (npx_on && A.R == 2) {
        while (npol - ip0 >= 3) {
            const int left = npol - ip0;
            const int nb = (left == 3 || left == 6 || left == 5) ? 3 : 4;      // 5 = 3 + 2, 6 = 3 + 3, 7 = 4 + 3, 9 = 4 + 3 + 2 ...
            if (nb == 4)
                hipLaunchKernelGGL((k_leg2_synth_npx<1, 4>), dim3((ntasks / 4) * 2), dim3(256), 0, s, A, tasks, ntasks, st, npol, ip0, ph, ph_stride, kq0 + 2 * ip0, 2);
            else
                hipLaunchKernelGGL((k_leg2_synth_npx<1, 3>), dim3((ntasks / 4) * 2), dim3(256), 0, s, A, tasks, ntasks, st, npol, ip0, ph, ph_stride, kq0 + 2 * ip0, 2);
            ip0 += nb;
        }
    }
    for (int ip = ip0; ip < npol; ++ip) {
        if (A.R == 2 && ip + 1 < npol && leg2_pairs2(false)) {
            hipLaunchKernelGGL(k_leg2_synth_np2<2>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, st, npol, ip, ph, ph_stride, kq0 + 2 * ip);
            ++ip;
            continue;
        }
        if (A.R == 4) { hipLaunchKernelGGL(k_leg2_synth<4>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, st, npol, ip, ph, ph_stride, kq0 + 2 * ip); continue; }
        if (A.R == 1) hipLaunchKernelGGL(k_leg2_synth<1>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, st, npol, ip, ph, ph_stride, kq0 + 2 * ip);
        else hipLaunchKernelGGL(k_leg2_synth<2>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, st, npol, ip, ph, ph_stride, kq0 + 2 * ip);
    }
}
void launch_leg2_adj(const Leg2Args& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                     int kq0, double* part, int64_t part_pol_stride, int64_t part_chunk_stride, int npol,
                     hipStream_t s) {
    if (ntasks == 0) return;
    // three and more polarisation pairs: four at a time on the matrix unit (k_leg2_adj_mx; CMDR_ADJ2_MX sets the smallest
    // batch that goes there, 0 disables); the rest -- and plans with one ring pair per lane -- through the VALU kernels
    const int mx_min = [] { const char* e = std::getenv("CMDR_ADJ2_MX"); return e ? std::atoi(e) : 3; }();   // per call (test hook)
    int ip0 = 0;
    if (A.R == 2 && mx_min > 0)
        while (npol - ip0 >= mx_min) {
            const int nb = std::min(4, npol - ip0);
            hipLaunchKernelGGL(k_leg2_adj_mx<1>, dim3(ntasks), dim3(128), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip0, nb,
                               part + ip0 * part_pol_stride, part_pol_stride, part_chunk_stride);
            ip0 += nb;
        }
    // one or two pairs left: the DPP form of the same task is available (CMDR_ADJ2_DX=1) but OFF by default: measured at
    // Nside 2048 / lmax 4000, one pair, 23.0 ms against 19.7 ms for k_leg2_adj -- two tile round trips per 32 l (mu+ and
    // mu-: the LDS store path, ~85 B/clk/CU, carries 16 B per (ring pair, l)) at two waves per SIMD cost more than the
    // wave-wide reductions they replace
    // CMDR_ADJ2_DX: 0 (default) the VALU kernels, 1 k_leg2_adj_dx, 2 its software-pipelined single-wave form k_leg2_adj_px.
    // Measured at Nside 2048 / lmax 4000, one pair: 19.9 ms (VALU), 23.0 ms (dx), 22.1 ms (px: 30.7 ms as first written,
    // 24.7 ms without fused ds_read2 / ds_write2 and with reads three blocks ahead, 22.1 ms as one wave per task without
    // workgroup barriers).  At one wave per SIMD the VALU is busy 44 % and the LDS array 57 % of the time, one after the
    // other rather than beside each other; the transposed forms stay opt-in.
    const int dx_mode = [] { const char* e = std::getenv("CMDR_ADJ2_DX"); return e ? std::atoi(e) : 0; }();
    const bool dx_on = dx_mode == 1;
    if (dx_mode == 2 && A.R == 2 && npol - ip0 >= 1 && npol - ip0 <= 2) {
        if (npol - ip0 == 2)
            hipLaunchKernelGGL((k_leg2_adj_px<2>), dim3(ntasks), dim3(64), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip0,
                               part + ip0 * part_pol_stride, part_pol_stride, part_chunk_stride);
        else
            hipLaunchKernelGGL((k_leg2_adj_px<1>), dim3(ntasks), dim3(64), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip0,
                               part + ip0 * part_pol_stride, part_pol_stride, part_chunk_stride);
        return;
    }
    if (dx_on && A.R == 2 && npol - ip0 >= 1 && npol - ip0 <= 2) {
        if (npol - ip0 == 2)
            hipLaunchKernelGGL((k_leg2_adj_dx<1, 2>), dim3(ntasks), dim3(128), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip0,
                               part + ip0 * part_pol_stride, part_pol_stride, part_chunk_stride);
        else
            hipLaunchKernelGGL((k_leg2_adj_dx<1, 1>), dim3(ntasks), dim3(128), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip0,
                               part + ip0 * part_pol_stride, part_pol_stride, part_chunk_stride);
        return;
    }
    for (int ip = ip0; ip < npol; ++ip) {
        if (A.R == 2 && ip + 1 < npol && leg2_pairs2(true)) {
            hipLaunchKernelGGL(k_leg2_adj_np2<2>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip, part + ip * part_pol_stride, part_pol_stride, part_chunk_stride);
            ++ip;
            continue;
        }
        if (A.R == 4) { hipLaunchKernelGGL(k_leg2_adj<4>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip, part + ip * part_pol_stride, part_chunk_stride); continue; }
        if (A.R == 1) hipLaunchKernelGGL(k_leg2_adj<1>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip, part + ip * part_pol_stride, part_chunk_stride);
        else hipLaunchKernelGGL(k_leg2_adj<2>, dim3(ntasks / 4), dim3(256), 0, s, A, tasks, ntasks, ph, ph_stride, kq0 + 2 * ip, part + ip * part_pol_stride, part_chunk_stride);
    }
}

__global__ void k_alm2_to_stream(const double* __restrict__ aE, const double* __restrict__ aB, int64_t pol_stride,
                                 double* __restrict__ st, int npol, const double* __restrict__ cnorm, int lmax) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax + 1) return;
    alm2_to_stream_elem(aE + blockIdx.z * pol_stride, aB + blockIdx.z * pol_stride, st, npol, blockIdx.z, cnorm, lmax, m, l);
}
void launch_alm2_to_stream(const double* aE, const double* aB, int64_t pol_stride, double* st, int npol,
                           const double* cnorm, int lmax, hipStream_t s) {
    dim3 grid((lmax + 2 + 255) / 256, lmax + 1, npol);
    hipLaunchKernelGGL(k_alm2_to_stream, grid, dim3(256), 0, s, aE, aB, pol_stride, st, npol, cnorm, lmax);
}
__global__ void k_part2_to_alm(const double* __restrict__ part, int64_t part_pol_stride, int64_t pcs, int nchunk,
                               double* __restrict__ aE, double* __restrict__ aB, int64_t pol_stride,
                               const double* __restrict__ cnorm, int lmax, const int* __restrict__ lwtab) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax) return;
    part2_to_alm_elem(part + blockIdx.z * part_pol_stride, pcs, nchunk, aE + blockIdx.z * pol_stride,
                      aB + blockIdx.z * pol_stride, cnorm, lmax, m, l, lwtab);
}
void launch_part2_to_alm(const double* part, int64_t part_pol_stride, int64_t pcs, int nchunk, double* aE, double* aB,
                         int64_t pol_stride, const double* cnorm, int lmax, int npol, hipStream_t s, const int* lwtab) {
    dim3 grid((lmax + 1 + 255) / 256, lmax + 1, npol);
    hipLaunchKernelGGL(k_part2_to_alm, grid, dim3(256), 0, s, part, part_pol_stride, pcs, nchunk, aE, aB, pol_stride,
                       cnorm, lmax, lwtab);
}

// ===================================================================================== ring stage
// One workgroup = one ring pair of one map; the packed complex spectrum / pixels live in dynamic LDS.
//   MODE 0: phases -> pixels (alm2map tail)         out map = y * (mul ? mul[pix] : 1) * (weighted ? wgt : 1)
//   MODE 1: pixels -> phases (map2alm head)         in  map * (mul ? mul[pix] : 1) * (weighted ? wgt : 1)
//   MODE 2: phases -> pixels * mul[pix] -> phases   (fused Y, N^-1, Y^T of the CR matvec; map never hits HBM)
template <int MODE>
__global__ void __launch_bounds__(1024) k_ring(const RingDev* __restrict__ rings, const int* __restrict__ cls,
                                              double* __restrict__ ph, int64_t ph_stride, int64_t prow /* rows (m) per pair of the phase layout */,
                                              double* __restrict__ map, int64_t map_stride,
                                              const double* const* __restrict__ mul, int weighted,
                                              const cd* __restrict__ tw, int log2Mmax,
                                              const cd* __restrict__ chirp, cd* __restrict__ scratch,
                                              int64_t scratch_map_stride, int scratch_line, int ncls, int nmaps,
                                              int per, int xbl, const cd* __restrict__ that, int64_t that_stride,
                                              int tw_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cd* buf = reinterpret_cast<cd*>(smem);
    // Workgroups with equal blockIdx % 8 share an XCD (and its L2).  Deal the class's ring pairs to the 8 groups in
    // blocks of 2^xbl neighbours (4 pairs x 32 B = one 128-B line of the phase array) and let each group walk block-major,
    // then map, then the pairs of the block: the workgroups that share the lines of a phase array (the same map, adjacent
    // pairs) are dispatched back to back on one XCD, so the first one's misses fill L2 for the others -- the phase loads
    // and stores are what this kernel spends its time on (65 KB between the 32-B entries of consecutive m of one pair;
    // timing experiments) -- and cheap (belt) and expensive (cap) rings spread evenly over the XCDs.
    const int grp = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int xb = 1 << xbl, blk = slot / (xb * nmaps), rem = slot - blk * xb * nmaps;
    const int imap = rem >> xbl, pl = blk * xb + (rem & (xb - 1));
    const int idx = (((pl >> xbl) * 8 + grp) << xbl) + (pl & ((1 << xbl) - 1));
    if (pl >= per || idx >= ncls) return;
    const int pair = cls[idx];
    const RingDev d = rings[pair];
    FftCtx c{(int)threadIdx.x, (int)blockDim.x};
    c.dbg = tw_off >> 24;
    tw_off &= (1 << 24) - 1;
    if (tw_off) ring_tw_fill(buf + tw_off, tw, log2Mmax, c);   // visible after the first block barrier (before any FFT pass)
    cd* sc = d.split ? scratch + imap * scratch_map_stride + (int64_t)(d.split - 1) * scratch_line : nullptr;
    ring_block<MODE>(buf, d, pair, ph + imap * ph_stride, prow, map ? map + imap * map_stride : nullptr,
                     mul ? mul[imap] : nullptr, weighted ? d.wgt : 1.0, tw, log2Mmax, chirp, sc, c,
                     that ? that + imap * that_stride : nullptr);
}

__global__ void __launch_bounds__(1024) k_ring_toeplitz_spec(const RingDev* __restrict__ rings, const int* __restrict__ cls,
                                                            const double* __restrict__ td, int64_t prow,
                                                            cd* __restrict__ that, const cd* __restrict__ tw,
                                                            int log2Mmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pair = cls[blockIdx.x];
    const RingDev d = rings[pair];
    ring_toeplitz_spec(reinterpret_cast<cd*>(smem), d, td, prow, pair, that, tw, log2Mmax,
                       FftCtx{(int)threadIdx.x, (int)blockDim.x});
}
void launch_ring_toeplitz_spec(const RingDev* rings, const int* cls, int ncls, int log2M, const double* td,
                               int64_t prow, cd* that, const cd* tw, int log2Mmax, hipStream_t s) {
    if (ncls == 0) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring_toeplitz_spec),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int nthr = std::max(64, std::min(512, (1 << log2M) / 2));
    hipLaunchKernelGGL(k_ring_toeplitz_spec, dim3(ncls), dim3(nthr), sizeof(cd) * (size_t)lds_elems(log2M), s, rings, cls, td,
                       prow, that, tw, log2Mmax);
}

void launch_ring(int mode, const RingDev* rings, const int* cls, int ncls, int log2M, double* ph,
                 int64_t ph_stride, int64_t prow /* rows (m) per pair of the phase layout */, double* map, int64_t map_stride, const double* const* mul,
                 int weighted, const cd* tw, int log2Mmax, const cd* chirp, cd* scratch, int64_t scratch_map_stride,
                 int scratch_line, int nmaps, hipStream_t s, const cd* that, int64_t that_stride) {
    if (ncls == 0 || nmaps == 0) return;
    static const int use_ldstw = [] { const char* e = std::getenv("CMDR_RING_LDSTW"); return e ? std::atoi(e) : 1; }();
    static const int dbg_skip = [] { const char* e = std::getenv("CMDR_RING_DEBUG_SKIP"); return e ? std::atoi(e) : 0; }();
    const int tw_off = (use_ldstw ? lds_elems(log2M) : 0) | (dbg_skip << 24);   // two-level twiddle table behind the FFT image
    const size_t lds = sizeof(cd) * (size_t)(lds_elems(log2M) + (use_ldstw ? ring_tw_elems(log2Mmax) : 0));
    int nthr = 512;   // measured: 512 > 256 threads per ring pair (more waves to cover LDS / global latency)
    if (const char* e = std::getenv("CMDR_RING_THREADS")) { const int v = std::atoi(e); if (v == 256 || v == 512 || v == 1024) nthr = v; }
    if (log2M >= 13)     // the 8192-point images (one workgroup per CU): their own knob
        if (const char* e = std::getenv("CMDR_RING_THREADS_BIG")) { const int v = std::atoi(e); if (v == 256 || v == 512 || v == 1024) nthr = v; }
    if (nthr > (1 << log2M) / 2) nthr = std::max(64, (1 << log2M) / 2);
    static int xbl = -1;                      // log2 of the pairs per block
    if (xbl < 0) { xbl = 2; if (const char* e = std::getenv("CMDR_RING_XBL")) { const int v = std::atoi(e); if (v >= 0 && v <= 10) xbl = v; } }
    const int xb = 1 << xbl;
    const int per = ((ncls + 8 * xb - 1) / (8 * xb)) * xb;   // pairs per XCD group: whole blocks
    dim3 grid(8 * per * nmaps);
#define CMDR_RING(MM)                                                                                            \
    do {                                                                                                         \
        static bool attr_set = false;                                                                            \
        if (!attr_set) {                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ring<MM>),                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                   \
            attr_set = true;                                                                                     \
        }                                                                                                        \
        hipLaunchKernelGGL(k_ring<MM>, grid, dim3(nthr), lds, s, rings, cls, ph, ph_stride, prow, map,      \
                           map_stride, mul, weighted, tw, log2Mmax, chirp, scratch, scratch_map_stride,          \
                           scratch_line, ncls, nmaps, per, xbl, that, that_stride, tw_off);                      \
    } while (0)
    if (mode == 0) CMDR_RING(0);
    else if (mode == 1) CMDR_RING(1);
    else CMDR_RING(2);
#undef CMDR_RING
}

// Masked monopole / dipole sums of applyMonoDipolePrior: one workgroup per ring pair, every thread walks its pixels of the
// northern and the southern ring, the 256 partial sums fold in a fixed tree -> out[pair][16] (deterministic; the host
// adds the pairs in order).
__global__ void __launch_bounds__(256) k_md_sums(const RingDev* __restrict__ rings, int nside, const double* __restrict__ map,
                                                const double* __restrict__ mask, int type, double* __restrict__ out) {
    __shared__ double red[256];
    const RingDev d = rings[blockIdx.x];
    double acc[kMdSums];
#pragma unroll
    for (int k = 0; k < kMdSums; ++k) acc[k] = 0.0;
    const double z = healpix_ring_z(nside, d.ring), sth = sqrt((1.0 - z) * (1.0 + z));
    const double dphi = 6.283185307179586476925287 / d.nphi;
    for (int k = threadIdx.x; k < d.nphi; k += 256) {
        const double phi = d.phi0 + dphi * k;
        md_pixel_accum(type, z, sth, phi, map[d.startN + k], mask[d.startN + k], acc);
        if (d.startS >= 0) md_pixel_accum(type, -z, sth, phi, map[d.startS + k], mask[d.startS + k], acc);
    }
    const int nsum = type == 1 ? 2 : 14;
    for (int q = 0; q < nsum; ++q) {
        red[threadIdx.x] = acc[q];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(int64_t)blockIdx.x * kMdSums + q] = red[0];
        __syncthreads();
    }
    if ((int)threadIdx.x >= nsum && (int)threadIdx.x < kMdSums) out[(int64_t)blockIdx.x * kMdSums + threadIdx.x] = 0.0;
}
void launch_md_sums(const RingDev* rings, int npair, int nside, const double* map, const double* mask, int type,
                    double* out, hipStream_t s) {
    if (npair == 0) return;
    hipLaunchKernelGGL(k_md_sums, dim3(npair), dim3(256), 0, s, rings, nside, map, mask, type, out);
}

// ===================================================================================== a_lm streaming kernels
// All run on an (l, m) grid: blockIdx.y = m, l = m + blockIdx.x*256 + threadIdx.x.

// Commander real-packed a_lm (one column) -> padded-triangle complex stream, times cnorm * kappa_m * scale.
// kappa_m = 1/sqrt2 for m>0 (Hermitian pair construction in the ring stage), 1 for m=0.
__global__ void k_alm_to_stream(const double* __restrict__ alm, int64_t alm_stride, double* __restrict__ ast,
                                int nbs, const double* __restrict__ cnorm, int lmax) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax + 1) return;
    for (int k = 0; k < nbs; ++k)   // the nbs entries of a stream row are contiguous: coalesced across l
        alm_to_stream_elem(alm + k * alm_stride, ast, nbs, k, cnorm, lmax, m, l);
}
void launch_alm_to_stream(const double* alm, int64_t alm_stride, double* ast, const double* cnorm, int lmax,
                          int nmaps, hipStream_t s) {
    dim3 grid((lmax + 2 + 255) / 256, lmax + 1);
    hipLaunchKernelGGL(k_alm_to_stream, grid, dim3(256), 0, s, alm, alm_stride, ast, nmaps, cnorm, lmax);
}

// partial columns -> Commander real-packed a_lm: alm = kappa'_m * cnorm * sum_chunks part ; kappa' = sqrt2 (m>0).
__global__ void k_part_to_alm(const double* __restrict__ part, int64_t part_map_stride, int64_t part_chunk_stride,
                              int nchunk, double* __restrict__ alm, int64_t alm_stride,
                              const double* __restrict__ cnorm, int lmax, const int* __restrict__ lwtab) {
    const int m = blockIdx.y, l = m + blockIdx.x * 256 + threadIdx.x;
    if (l > lmax) return;
    part_to_alm_elem(part + blockIdx.z * part_map_stride, part_chunk_stride, nchunk, alm + blockIdx.z * alm_stride,
                     cnorm, lmax, m, l, lwtab);
}
void launch_part_to_alm(const double* part, int64_t pms, int64_t pcs, int nchunk, double* alm, int64_t alm_stride,
                        const double* cnorm, int lmax, int nmaps, hipStream_t s, const int* lwtab) {
    dim3 grid((lmax + 1 + 255) / 256, lmax + 1, nmaps);
    hipLaunchKernelGGL(k_part_to_alm, grid, dim3(256), 0, s, part, pms, pcs, nchunk, alm, alm_stride, cnorm, lmax, lwtab);
}

}  // namespace cmdr
