// Chain-file I/O of the amplitude sample: what comm_diffuse_comp%dumpFITS / comm_map%writeFITS write
// into the HDF5 chain file per Gibbs iteration and component, and what initDiffuseHDF / comm_Cl%initHDF read back on a
// restart (commander3/src/comm_diffuse_comp_mod.f90:2459-2524, 2687-2728; comm_map_mod.f90:712-745, 860-889;
// comm_Cl_mod.f90:1335, 1393):
//   /<iter, 6 digits>/<label>/amp_alm    float32, Fortran shape ((lmax+1)^2, nmaps), row index l^2 + l + m
//   /<iter>/<label>/amp_lmax, amp_nmaps  int32 scalars
//   /<iter>/<label>/sigma_l              float64, Fortran shape (0:lmax, nspec)        (getSigmaL x RJ2unit)
//   /<iter>/<label>/Dl                   float64, Fortran shape (0:lmax, nspec)        (cltype 'binned')
// The HDF5 Fortran API stores a Fortran array of shape (n1, n2) as a C-order dataset of dims [n2][n1].
// HDF5 is bound at run time (dlopen of libhdf5: the image ships 1.10.6 under /opt/conda/lib), so the library has no
// link-time dependency on it.  Host-side formatting only: O(nalm) index shuffles, no transforms, nothing of the solve.
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cmdr_hip.h"
#include "common_host.hpp"
#include "geom.hpp"

namespace cmdr {
namespace {

using hid = int64_t;            // hid_t of HDF5 >= 1.10
using hsz = unsigned long long; // hsize_t

struct H5 {
    void* h = nullptr;
    std::string err;
    int (*open)() = nullptr;
    int (*get_libversion)(unsigned*, unsigned*, unsigned*) = nullptr;
    int (*Eset_auto2)(hid, void*, void*) = nullptr;
    hid (*Fcreate)(const char*, unsigned, hid, hid) = nullptr;
    hid (*Fopen)(const char*, unsigned, hid) = nullptr;
    int (*Fclose)(hid) = nullptr;
    int (*Lexists)(hid, const char*, hid) = nullptr;
    int (*Ldelete)(hid, const char*, hid) = nullptr;
    hid (*Gcreate2)(hid, const char*, hid, hid, hid) = nullptr;
    int (*Gclose)(hid) = nullptr;
    hid (*Screate_simple)(int, const hsz*, const hsz*) = nullptr;
    hid (*Screate)(int) = nullptr;
    int (*Sclose)(hid) = nullptr;
    hid (*Dcreate2)(hid, const char*, hid, hid, hid, hid, hid) = nullptr;
    hid (*Dopen2)(hid, const char*, hid) = nullptr;
    int (*Dwrite)(hid, hid, hid, hid, hid, const void*) = nullptr;
    int (*Dread)(hid, hid, hid, hid, hid, void*) = nullptr;
    hid (*Dget_space)(hid) = nullptr;
    int (*Sget_simple_extent_ndims)(hid) = nullptr;
    int (*Sget_simple_extent_dims)(hid, hsz*, hsz*) = nullptr;
    int (*Dclose)(hid) = nullptr;
    hid f32 = -1, f64 = -1, i32 = -1;
};

H5& h5() {
    static H5 A;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"libhdf5.so.103", "libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so",
                               "libhdf5_serial.so.103", "libhdf5_serial.so"};
        // CMDR_HDF5_LIB names THE library to bind (no fall-back to the default names: a wrong path is an error)
        const char* forced = std::getenv("CMDR_HDF5_LIB");
        if (forced) A.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char* n : names) {
                if (A.h) break;
                A.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            }
        if (!A.h) {
            const char* e = dlerror();           // ONE call: dlerror() clears the message it returns
            A.err = std::string("cannot load libhdf5: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(A.h, n);
            if (!p && A.err.empty()) A.err = std::string("libhdf5 lacks ") + n;
            return p;
        };
#define CMDR_H5(field, name) A.field = (decltype(A.field))sym(name)
        CMDR_H5(open, "H5open"); CMDR_H5(get_libversion, "H5get_libversion"); CMDR_H5(Eset_auto2, "H5Eset_auto2");
        CMDR_H5(Fcreate, "H5Fcreate"); CMDR_H5(Fopen, "H5Fopen"); CMDR_H5(Fclose, "H5Fclose");
        CMDR_H5(Lexists, "H5Lexists"); CMDR_H5(Ldelete, "H5Ldelete"); CMDR_H5(Gcreate2, "H5Gcreate2"); CMDR_H5(Gclose, "H5Gclose");
        CMDR_H5(Screate_simple, "H5Screate_simple"); CMDR_H5(Screate, "H5Screate"); CMDR_H5(Sclose, "H5Sclose");
        CMDR_H5(Dcreate2, "H5Dcreate2"); CMDR_H5(Dopen2, "H5Dopen2"); CMDR_H5(Dwrite, "H5Dwrite"); CMDR_H5(Dread, "H5Dread");
        CMDR_H5(Dget_space, "H5Dget_space"); CMDR_H5(Sget_simple_extent_ndims, "H5Sget_simple_extent_ndims");
        CMDR_H5(Sget_simple_extent_dims, "H5Sget_simple_extent_dims"); CMDR_H5(Dclose, "H5Dclose");
#undef CMDR_H5
        if (!A.err.empty()) return;
        unsigned a = 0, b = 0, c = 0;
        if (A.open() < 0 || A.get_libversion(&a, &b, &c) < 0) { A.err = "H5open failed"; return; }
        if (a != 1 || b < 10) { A.err = "libhdf5 " + std::to_string(a) + "." + std::to_string(b) + ": need 1.10 or later (64-bit hid_t)"; return; }
        hid* p;
        if ((p = (hid*)sym("H5T_NATIVE_FLOAT_g"))) A.f32 = *p;
        if ((p = (hid*)sym("H5T_NATIVE_DOUBLE_g"))) A.f64 = *p;
        if ((p = (hid*)sym("H5T_NATIVE_INT_g"))) A.i32 = *p;
        (void)A.Eset_auto2(0, nullptr, nullptr);    // errors come back as return codes, reported through cmdr_last_error
    });
    if (!A.err.empty()) throw Error("HDF5 unavailable: " + A.err);
    return A;
}

struct File {
    hid id = -1;
    ~File() { if (id >= 0) (void)h5().Fclose(id); }
};

void ensure_group(hid f, const std::string& path) {
    H5& A = h5();
    if (A.Lexists(f, path.c_str(), 0) > 0) return;
    const hid g = A.Gcreate2(f, path.c_str(), 0, 0, 0);
    CMDR_REQUIRE(g >= 0, "cannot create HDF5 group");
    (void)A.Gclose(g);
}

void write_ds(hid f, const std::string& path, hid type, int rank, const hsz* dims, const void* data) {
    H5& A = h5();
    if (A.Lexists(f, path.c_str(), 0) > 0) CMDR_REQUIRE(A.Ldelete(f, path.c_str(), 0) >= 0, "cannot replace HDF5 dataset");
    const hid sp = rank == 0 ? A.Screate(0 /*H5S_SCALAR*/) : A.Screate_simple(rank, dims, nullptr);
    CMDR_REQUIRE(sp >= 0, "H5Screate failed");
    const hid d = A.Dcreate2(f, path.c_str(), type, sp, 0, 0, 0);
    if (d < 0) { (void)A.Sclose(sp); throw Error("cannot create HDF5 dataset " + path); }
    const int rc = A.Dwrite(d, type, 0, 0, 0, data);
    (void)A.Dclose(d);
    (void)A.Sclose(sp);
    CMDR_REQUIRE(rc >= 0, "H5Dwrite failed");
}

// reads a dataset of exactly the given dims (any stored numeric type; HDF5 converts to memtype)
void read_ds(hid f, const std::string& path, hid memtype, int rank, const hsz* dims, void* data) {
    H5& A = h5();
    CMDR_REQUIRE(A.Lexists(f, path.c_str(), 0) > 0, ("chain file has no dataset " + path).c_str());
    const hid d = A.Dopen2(f, path.c_str(), 0);
    CMDR_REQUIRE(d >= 0, "H5Dopen failed");
    const hid sp = A.Dget_space(d);
    hsz got[4] = {0, 0, 0, 0};
    const int nr = A.Sget_simple_extent_ndims(sp);
    bool ok = nr == rank;
    if (ok && rank > 0) { (void)A.Sget_simple_extent_dims(sp, got, nullptr); for (int i = 0; i < rank; ++i) ok = ok && got[i] == dims[i]; }
    (void)A.Sclose(sp);
    if (!ok) { (void)A.Dclose(d); throw Error("dataset " + path + " does not have the expected shape"); }
    const int rc = A.Dread(d, memtype, 0, 0, 0, data);
    (void)A.Dclose(d);
    CMDR_REQUIRE(rc >= 0, "H5Dread failed");
}

std::string comp_path(int iter, const char* label) {
    char it[16];
    std::snprintf(it, sizeof(it), "%06d", iter);           // int2string(iter, itext), itext = character(len=6)
    return std::string("/") + it + "/" + label;
}

}  // namespace
}  // namespace cmdr

using namespace cmdr;

extern "C" {

int cmdr_chain_write_comp(const char* chainfile, int iter, const char* label, const double* alm, int lmax, int nmaps,
                          const double* unit_scale, const double* sigma_l, const double* Dl) {
    try {
        CMDR_REQUIRE(chainfile && label && alm && lmax >= 0 && nmaps >= 1 && nmaps <= 3 && iter >= 0 && iter <= 999999, "bad arguments");
        H5& A = h5();
        File F;
        F.id = A.Fopen(chainfile, 1 /*H5F_ACC_RDWR*/, 0);
        if (F.id < 0) F.id = A.Fcreate(chainfile, 2 /*H5F_ACC_TRUNC*/, 0, 0);
        CMDR_REQUIRE(F.id >= 0, "cannot open or create the chain file");
        const std::string grp = comp_path(iter, label);
        ensure_group(F.id, grp.substr(0, 7));
        ensure_group(F.id, grp);
        const int64_t na = nalm_packed(lmax);
        std::vector<float> c32((size_t)na * nmaps);                     // [nmaps][l^2 + l + m]  (comm_map_mod.f90:712-740)
        for (int k = 0; k < nmaps; ++k) {
            const double sc = unit_scale ? unit_scale[k] : 1.0;          // RJ2unit_ * cg_scale (comm_diffuse_comp_mod.f90:2459-2461)
            for (int m = 0; m <= lmax; ++m)
                for (int l = m; l <= lmax; ++l) {
                    const int64_t i = mind(lmax, m) + (m == 0 ? l : 2 * (l - m));
                    c32[(size_t)k * na + (int64_t)l * l + l + m] = (float)(alm[(size_t)k * na + i] * sc);
                    if (m > 0) c32[(size_t)k * na + (int64_t)l * l + l - m] = (float)(alm[(size_t)k * na + i + 1] * sc);
                }
        }
        const hsz d2[2] = {(hsz)nmaps, (hsz)na};
        write_ds(F.id, grp + "/amp_alm", A.f32, 2, d2, c32.data());
        write_ds(F.id, grp + "/amp_lmax", A.i32, 0, nullptr, &lmax);
        write_ds(F.id, grp + "/amp_nmaps", A.i32, 0, nullptr, &nmaps);
        const int nspec = nmaps * (nmaps + 1) / 2;
        const hsz ds[2] = {(hsz)nspec, (hsz)(lmax + 1)};
        if (sigma_l) write_ds(F.id, grp + "/sigma_l", A.f64, 2, ds, sigma_l);
        if (Dl) write_ds(F.id, grp + "/Dl", A.f64, 2, ds, Dl);
        return 0;
    } catch (const std::exception& e) {
        cmdr::set_last_error(e.what());
        return -1;
    }
}

int cmdr_chain_read_comp(const char* chainfile, int iter, const char* label, int lmax, int nmaps,
                         const double* unit_scale, double* alm, double* Dl) {
    try {
        CMDR_REQUIRE(chainfile && label && alm && lmax >= 0 && nmaps >= 1 && nmaps <= 3, "bad arguments");
        H5& A = h5();
        File F;
        F.id = A.Fopen(chainfile, 0 /*H5F_ACC_RDONLY*/, 0);
        CMDR_REQUIRE(F.id >= 0, "cannot open the chain file");
        const std::string grp = comp_path(iter, label);
        const int64_t na = nalm_packed(lmax);
        std::vector<double> c((size_t)na * nmaps);
        const hsz d2[2] = {(hsz)nmaps, (hsz)na};
        read_ds(F.id, grp + "/amp_alm", A.f64, 2, d2, c.data());           // stored float32, widened by HDF5 (readHDF :882)
        for (int k = 0; k < nmaps; ++k) {
            const double sc = unit_scale ? unit_scale[k] : 1.0;            // initDiffuseHDF :2722-2724 (a division there too)
            for (int m = 0; m <= lmax; ++m)
                for (int l = m; l <= lmax; ++l) {
                    const int64_t i = mind(lmax, m) + (m == 0 ? l : 2 * (l - m));
                    alm[(size_t)k * na + i] = c[(size_t)k * na + (int64_t)l * l + l + m] / sc;
                    if (m > 0) alm[(size_t)k * na + i + 1] = c[(size_t)k * na + (int64_t)l * l + l - m] / sc;
                }
        }
        if (Dl) {
            const int nspec = nmaps * (nmaps + 1) / 2;
            const hsz ds[2] = {(hsz)nspec, (hsz)(lmax + 1)};
            read_ds(F.id, grp + "/Dl", A.f64, 2, ds, Dl);
        }
        return 0;
    } catch (const std::exception& e) {
        cmdr::set_last_error(e.what());
        return -1;
    }
}

}  // extern "C"
