// HEALPix RING geometry and Commander's a_lm index maps (host side).
//
// Geometry: published HEALPix RING scheme.  Commander builds its pixel ownership with
// HEALPix in_ring and sorts it (commander3/src/comm_map_mod.f90:193-226): rank r of P owns rings
// i = 1+r, 1+r+P, ... <= 2*nside and the mirror 4*nside-i; the local map is those rings in ascending ring order.
// a_lm layout: m-major real-packed (commander3/src/comm_map_mod.f90:228-261, lm2i :1213-1246).
#pragma once
#include <cmath>
#include <cstdint>

namespace cmdr {

constexpr double kPi = 3.141592653589793238462643383279502884;

struct RingInfo {
    int nphi;       // pixels in ring
    double z;       // cos(theta)
    double sth;     // sin(theta)
    double phi0;    // azimuth of first pixel
    int64_t start;  // full-sky RING index of first pixel
};

// ring = 1..4*nside-1, north to south
inline RingInfo healpix_ring(int nside, int ring) {
    RingInfo r;
    const int64_t N = nside, npix = 12 * N * N;
    const int nr = ring > 2 * nside ? 4 * nside - ring : ring;
    const double fN = (double)N;
    if (nr < nside) {
        const double i = (double)nr;
        const double omz = i * i / (3.0 * fN * fN);
        r.z = 1.0 - omz;
        r.sth = std::sqrt(omz * (2.0 - omz));
        r.nphi = 4 * nr;
        r.phi0 = kPi / (4.0 * i);
        r.start = 2 * (int64_t)nr * (nr - 1);
    } else {
        r.z = 4.0 / 3.0 - 2.0 * (double)nr / (3.0 * fN);
        r.sth = std::sqrt((1.0 - r.z) * (1.0 + r.z));
        r.nphi = 4 * nside;
        r.phi0 = ((nr - nside) & 1) ? 0.0 : kPi / (4.0 * fN);
        r.start = 2 * N * (N - 1) + 4 * N * (int64_t)(nr - nside);
    }
    if (ring != nr) {
        r.z = -r.z;
        r.start = npix - r.start - r.nphi;
    }
    return r;
}

// Complex (l,m) triangle, m-major: index of (l=m, m); entries l=m..lmax follow contiguously.
inline int64_t moff(int lmax, int m) { return (int64_t)m * (lmax + 1) - (int64_t)m * (m - 1) / 2; }
inline int64_t ntri(int lmax) { return (int64_t)(lmax + 1) * (lmax + 2) / 2; }
// Commander's real-packed start of block m (P=1): m=0 -> 0 (lmax+1 reals), m>0 -> interleaved (+m,-m) pairs.
inline int64_t mind(int lmax, int m) { return m == 0 ? 0 : 2 * moff(lmax, m) - (lmax + 1); }
inline int64_t nalm_packed(int lmax) { return (int64_t)(lmax + 1) * (lmax + 1); }

// libsharp's (ring,m) cut for spin 0: lambda_lm(theta) is negligible for every l <= lmax when m exceeds this.
inline int mlim_spin0(int lmax, double sth) {
    double ofs = lmax * 0.01;
    if (ofs < 100.) ofs = 100.;
    double res = lmax * sth + ofs;
    if (res > lmax) res = lmax;
    return (int)(res + 0.5);
}

}  // namespace cmdr
