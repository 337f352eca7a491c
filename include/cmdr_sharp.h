/*
 * cmdr_sharp.h -- the literal libsharp2 symbols that commander3/src/sharp.f90 binds, exported by
 * libcmdr_hip.so so that Commander3's existing `module sharp` links against the GPU library unchanged.
 *
 * Supported subset = exactly what Commander issues on this path: real-packed m-major a_lm
 * (sharp_make_mmajor_real_packed_alm_info, sharp.f90:44-50,128), HEALPix ring subsets with optional ring weights
 * (sharp_make_subset_healpix_geom_info, sharp.f90:64-71,158), double precision (SHARP_DP always set, sharp.f90:206),
 * job types YtW/Y/Yt/WY (sharp.f90:8-14), one transform per call (sharp.f90:211-216), spin 0 (one column) and
 * spin 2 (alm = E,B ; map = Q,U ; comm_map_mod.f90:446-449).  SHARP_ADD is never set by
 * any caller and is rejected.  Like libsharp2 the functions return void; on an unsupported request or a HIP failure
 * they print the reason and abort() -- libsharp2's own convention ("library aborts internally").
 *
 * Distribution: libsharp2 redistributes a_lm (by m) and rings across the MPI ranks of `comm` inside
 * sharp_execute_mpi_fortran (ownership: comm_map_mod.f90:193-261 -- rank r of P owns rings r+1, r+1+P, ... with their
 * mirrors and m = r, r+P, ...).  Here every rank transforms ITS rings with ALL m on its GPU, so the exchange is one sum
 * of the full packed a_lm over the communicator (before the transform for Y / WY, after it for Yt / YtW).  The library
 * does not link MPI: the driver registers, per communicator handle, a routine that sums a host buffer of doubles over
 * that communicator (cmdr_sharp_register_comm; INTEGRATION.md shows the 6-line MPI_Allreduce wrapper).  An
 * unregistered communicator must be a one-rank chain (every m and ring local).  The CR-level API of cmdr_hip.h keeps
 * vectors resident and is the fast path; this level exists so that `module sharp` links unchanged.
 */
#ifndef CMDR_SHARP_H
#define CMDR_SHARP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sharp_alm_info sharp_alm_info;
typedef struct sharp_geom_info sharp_geom_info;

enum { SHARP_YtW = 0, SHARP_Y = 1, SHARP_Yt = 2, SHARP_WY = 3, SHARP_ALM2MAP_DERIV1 = 4 };   /* sharp.f90:8-14 */
enum { SHARP_DP = 1 << 4, SHARP_ADD = 1 << 5, SHARP_REAL_HARMONICS = 1 << 6, SHARP_NO_FFT = 1 << 7 }; /* :17-20 */

/* sharp.f90:44-50 (called with stride = 1, :128) */
void sharp_make_mmajor_real_packed_alm_info(int lmax, int stride, int nm, const int* ms, sharp_alm_info** alm_info);
/* sharp.f90:35-42 (declared by Commander, never called): accepted only for the packed real m-major case */
void sharp_make_general_alm_info(int lmax, int nm, int stride, const int* mval, const ptrdiff_t* mvstart, int flags,
                                 sharp_alm_info** alm_info);
ptrdiff_t sharp_alm_count(const sharp_alm_info* self);                        /* sharp.f90:52-56 */
void sharp_destroy_alm_info(sharp_alm_info* info);                            /* sharp.f90:58-61 */
/* sharp.f90:64-71: rings = 1-based ring numbers (NULL = all 4*nside-1), weight[2*nside] or NULL */
void sharp_make_subset_healpix_geom_info(int nside, int stride, int nrings, const int* rings, const double* weight,
                                         sharp_geom_info** geom_info);
ptrdiff_t sharp_map_size(const sharp_geom_info* info);                        /* sharp.f90:78-82 */
void sharp_destroy_geom_info(sharp_geom_info* info);                          /* sharp.f90:73-76 */
/* sharp.f90:86-94: alm / map = arrays of column pointers (void**) */
void sharp_execute(int type, int spin, void* alm, void* map, const sharp_geom_info* geom_info,
                   const sharp_alm_info* alm_info, int flags, double* time, unsigned long long* opcnt);
/* In-place sum of n doubles at a HOST address over the ranks of one communicator; comm = the Fortran handle
 * (MPI_Fint) the driver later passes to sharp_execute_mpi_fortran.  fn = NULL removes the registration. */
typedef void (*cmdr_sharp_allreduce_fn)(void* user, double* host_buf, long long n);
void cmdr_sharp_register_comm(int comm, cmdr_sharp_allreduce_fn fn, void* user);
/* sharp.f90:96-104: comm = Fortran MPI communicator handle (registered above, or a one-rank communicator) */
void sharp_execute_mpi_fortran(int comm, int type, int spin, void* alm, void* map, const sharp_geom_info* geom_info,
                               const sharp_alm_info* alm_info, int flags, double* time, unsigned long long* opcnt);

#ifdef __cplusplus
}
#endif
#endif
