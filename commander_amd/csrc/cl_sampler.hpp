// C_l Gibbs step, host side (see cl_sampler.cpp).
#pragma once
#include "../../include/cmdr_hip.h"

namespace cmdr {
// comm_Cl%updateS: returns the number of multipoles whose matrix was not positive definite (the reference leaves
// A(1,1) = -1e30 there and carries on; so does this).
int cl_update_S(int lmax, int nmaps, int lmin, const double* Dl, const double* RJ2unit, double* sqrtS, double* sqrtInvS,
                double* S);
// sample_Cls_inverse_wishart2 (no lookup): 0 = ok, 1 = the reference's ok = .false.
int cl_sample_binned(int lmax, int nmaps, const double* sigma_l, const double* S_mat, const double* RJ2unit, int nbin,
                     const cmdr_cl_bin* bins, const double* uniform, int nuniform, double* Dl, int* nused);
int cl_sample_lookup(int lmax, int lmin_lookup, int lmax_lookup, int nmodel, const double* Dl_lookup, const int* active,
                     const double* sigma_l, const double* S_mat, const double* RJ2unit, double uniform, double* Dl,
                     int* chosen);
double cl_apod(int l, int l_apod, int lmax, int lmax_prior, bool positive);
void cl_apply_apod(int lmax, int nmaps, int l_apod, int lmax_prior, double* sqrtS, double* sqrtInvS, double* S);
}  // namespace cmdr
