/*
 * cmdr_hip.h -- C ABI of libcmdr_hip.so: the MI355X-native replacement for the Gibbs amplitude-sampling path of
 * Commander3 (constrained-realization PCG solve + the spherical-harmonic transforms it calls).
 *
 * Two entry levels:
 *   (1) SHT level -- what commander3/src/sharp.f90 binds from libsharp2 today, with explicit sizes and an
 *       int status instead of void/abort (the literal libsharp2 symbol names are in include/cmdr_sharp.h);
 *   (2) CR level -- the bodies of cr_matmulA / cr_invM / cr_computeRHS / solve_cr_eqn_by_CG
 *       (commander3/src/comm_cr_mod.f90) with all vectors resident in HBM for the whole solve.
 *
 * Conventions: every function returns 0 on success, a negative value on error (message via cmdr_last_error());
 * all arrays are fp64 (Fortran real(dp) / c_double), column-major exactly as the Fortran side holds them;
 * "packed a_lm" is Commander's m-major real-packed layout (commander3/src/comm_map_mod.f90:228-261);
 * maps are HEALPix RING ordered, restricted to the rings the plan owns in ascending ring order
 * (commander3/src/comm_map_mod.f90:193-226).  One host thread per context; one context per (chain, GPU).
 * The library is HIP-only: it fails loudly (error code) when no gfx950 device is usable -- there is no CPU path.
 */
#ifndef CMDR_HIP_H
#define CMDR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmdr_sht_plan cmdr_sht_plan;
typedef struct cmdr_ctx cmdr_ctx;

/* job types: same values as commander3/src/sharp.f90:8-14 */
enum { CMDR_YtW = 0, CMDR_Y = 1, CMDR_Yt = 2, CMDR_WY = 3 };

const char* cmdr_last_error(void);
/* number of visible HIP devices (0 if none / runtime unusable) */
int cmdr_device_count(void);
int cmdr_set_device(int device);
int cmdr_device_synchronize(void);

/* ---- device memory helpers (so a host language without HIP bindings can keep data resident) ---- */
int cmdr_dev_alloc(size_t nbytes, void** out);
int cmdr_dev_free(void* p);
int cmdr_memcpy_h2d(void* dst_dev, const void* src_host, size_t nbytes);
int cmdr_memcpy_d2h(void* dst_host, const void* src_dev, size_t nbytes);

/* ------------------------------------------------------------------------------------------------
 * SHT level.  Replaces sharp_make_mmajor_real_packed_alm_info + sharp_make_subset_healpix_geom_info
 * (sharp.f90:44-71, called from comm_map_mod.f90:264-283) and sharp_execute (sharp.f90:86-104, called from
 * comm_map_mod.f90:437-579).
 *   rings   : nrings northern ring numbers (1..2*nside) owned by this plan; the mirror ring 4*nside-i is
 *             implied (comm_map_mod.f90:197-221).  NULL / 0 = all rings.
 *   wring   : [2*nside] ring weights W (Commander passes 1 + weight_ring); NULL = 1.
 *   max_maps: how many columns one call may transform at once (workspace is sized for it).
 */
int cmdr_sht_plan_create(int nside, int lmax, int nrings, const int* rings, const double* wring, int max_maps,
                         cmdr_sht_plan** out);
int cmdr_sht_plan_destroy(cmdr_sht_plan* plan);
int64_t cmdr_sht_nalm(const cmdr_sht_plan* plan);   /* sharp_alm_count  (sharp.f90:52-56) */
int64_t cmdr_sht_npix(const cmdr_sht_plan* plan);   /* sharp_map_size   (sharp.f90:78-82) */
/* One spin-0 transform per column; alm[k] / map[k] are the column pointers, as in sharp_execute (sharp.f90:203-224).
 * Host-pointer form copies in and out; the _dev form takes device pointers and leaves results in HBM. */
int cmdr_sht_execute(cmdr_sht_plan* plan, int job, int nmaps, double* const* alm, double* const* map);
int cmdr_sht_execute_dev(cmdr_sht_plan* plan, int job, int nmaps, double* alm_dev, int64_t alm_stride,
                         double* map_dev, int64_t map_stride);

#ifdef __cplusplus
}
#endif
#endif
