// Launchers of the gfx950 kernels (implemented in kernels.hip / cr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels_body.hpp"
#include "plan_tables.hpp"

namespace cmdr {

void launch_leg_synth(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ast, int64_t ast_stride,
                      double* ph, int64_t ph_stride, int nmaps, hipStream_t s);
void launch_leg_adj(const LegArgs& A, const WaveTask* tasks, int ntasks, const double* ph, int64_t ph_stride,
                    double* part, int64_t part_map_stride, int64_t part_chunk_stride, int nmaps, bool square,
                    hipStream_t s);
// mode 0: phases->map, 1: map->phases, 2: phases -> *mul -> phases (in place)
void launch_ring(int mode, const RingDev* rings, const int* cls, int ncls, int log2M, double* ph,
                 int64_t ph_stride, int64_t npair_pad, double* map, int64_t map_stride, const double* const* mul,
                 int weighted, const cd* tw, int log2Mmax, const cd* chirp, int nmaps, hipStream_t s);
void launch_alm_to_stream(const double* alm, int64_t alm_stride, double* ast, int64_t ast_stride,
                          const double* cnorm, int lmax, int nmaps, hipStream_t s);
void launch_part_to_alm(const double* part, int64_t pms, int64_t pcs, int nchunk, double* alm, int64_t alm_stride,
                        const double* cnorm, int lmax, int nmaps, hipStream_t s);

}  // namespace cmdr
