"""ctypes loader of libcmdr_hip.so (the C ABI in include/cmdr_hip.h)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CMDR_LIB_PATH") or os.path.join(_HERE, "libcmdr_hip.so")   # CMDR_LIB_PATH: A/B builds (development)
_LIB = None


class CmdrError(RuntimeError):
    pass


def build_library(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _SO


class ClBin(ctypes.Structure):
    """cmdr_cl_bin of include/cmdr_hip.h."""
    _fields_ = [("lmin", ctypes.c_int), ("lmax", ctypes.c_int), ("spec", ctypes.c_int), ("sample", ctypes.c_int),
                ("sigma", ctypes.c_double)]


def _sig(L):
    c_int, c_i64, c_sz, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    pdp = ctypes.POINTER(dp)
    L.cmdr_last_error.restype = ctypes.c_char_p
    L.cmdr_device_count.restype = c_int
    L.cmdr_set_device.argtypes = [c_int]
    L.cmdr_dev_alloc.argtypes = [c_sz, ctypes.POINTER(c_vp)]
    L.cmdr_dev_free.argtypes = [c_vp]
    L.cmdr_dev_mem_info.argtypes = [ctypes.POINTER(c_sz), ctypes.POINTER(c_sz)]
    L.cmdr_memcpy_h2d.argtypes = [c_vp, c_vp, c_sz]
    L.cmdr_memcpy_d2h.argtypes = [c_vp, c_vp, c_sz]
    L.cmdr_host_register.argtypes = [c_vp, c_sz]
    L.cmdr_host_unregister.argtypes = [c_vp]
    L.cmdr_sht_plan_create.argtypes = [c_int, c_int, c_int, ip, dp, c_int, ctypes.POINTER(c_vp)]
    L.cmdr_sht_plan_create_pol.argtypes = [c_int, c_int, c_int, ip, dp, c_int, ctypes.POINTER(c_vp)]
    L.cmdr_sht_execute_spin2.argtypes = [c_vp, c_int, dp, dp, dp, dp]
    L.cmdr_sht_execute_spin2_dev.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_vp]
    L.cmdr_sht_plan_destroy.argtypes = [c_vp]
    L.cmdr_sht_nalm.argtypes = [c_vp]
    L.cmdr_sht_nalm.restype = c_i64
    L.cmdr_sht_npix.argtypes = [c_vp]
    L.cmdr_sht_npix.restype = c_i64
    L.cmdr_sht_execute.argtypes = [c_vp, c_int, c_int, pdp, pdp]
    L.cmdr_sht_execute_dev.argtypes = [c_vp, c_int, c_int, c_vp, c_i64, c_vp, c_i64]
    pvp = ctypes.POINTER(c_vp)
    c_dbl = ctypes.c_double
    L.cmdr_ctx_create.argtypes = [c_int, pvp]
    L.cmdr_ctx_destroy.argtypes = [c_vp]
    L.cmdr_ctx_set_rings.argtypes = [c_vp, c_int, c_int, ip]
    L.cmdr_ctx_set_allreduce.argtypes = [c_vp, c_vp, c_vp]
    L.cmdr_ctx_set_allreduce_stream.argtypes = [c_vp, c_vp, c_vp]
    L.cmdr_ctx_set_band_sharding.argtypes = [c_vp, c_vp, c_vp, c_int]
    L.cmdr_ctx_set_only_pol.argtypes = [c_vp, c_int]
    L.cmdr_ctx_set_literal_quirks.argtypes = [c_vp, c_int]
    L.cmdr_rccl_unique_id.argtypes = [ctypes.c_char_p]
    L.cmdr_rccl_version.restype = c_int
    L.cmdr_ctx_init_rccl.argtypes = [c_vp, ctypes.c_char_p, c_int, c_int]
    L.cmdr_ctx_rccl_split_rings.argtypes = [c_vp, c_int, c_int, c_int]
    L.cmdr_ctx_drop_rccl.argtypes = [c_vp]
    L.cmdr_ctx_set_vector_slicing.argtypes = [c_vp, c_int, c_int]
    L.cmdr_ctx_rccl_size.argtypes = [c_vp]
    L.cmdr_ctx_rccl_size.restype = c_int
    L.cmdr_band_add.argtypes = [c_vp, c_int, c_int, c_int, dp, dp, c_dbl, dp, dp]
    L.cmdr_comp_add.argtypes = [c_vp, c_int, c_int, c_int, dp, dp, dp, dp, c_int]
    L.cmdr_finalize.argtypes = [c_vp]
    L.cmdr_ncr.argtypes = [c_vp]
    L.cmdr_ncr.restype = c_i64
    L.cmdr_band_npix.argtypes = [c_vp, c_int]
    L.cmdr_band_npix.restype = c_i64
    L.cmdr_precond_init_diag.argtypes = [c_vp]
    L.cmdr_precond_update_diag.argtypes = [c_vp]
    L.cmdr_get_invN_diag.argtypes = [c_vp, c_int, dp]
    L.cmdr_precond_set_lowl.argtypes = [c_vp, c_int, c_int, ip, pdp]
    L.cmdr_precond_init_pseudoinv.argtypes = [c_vp]
    L.cmdr_precond_update_pseudoinv.argtypes = [c_vp]
    L.cmdr_get_alpha_nu.argtypes = [c_vp, c_int, dp]
    L.cmdr_band_set_qucov.argtypes = [c_vp, c_int, dp, dp]
    L.cmdr_compact_add.argtypes = [c_vp, c_int, dp, dp, c_int]
    L.cmdr_compact_set_band.argtypes = [c_vp, c_int, c_int, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), dp]
    L.cmdr_comp_set_mixing_map.argtypes = [c_vp, c_int, c_int, dp, c_int]
    L.cmdr_comp_set_cl_diag.argtypes = [c_vp, c_int, dp]
    L.cmdr_comp_set_cl.argtypes = [c_vp, c_int, dp, dp, dp]
    L.cmdr_comp_set_f_mean.argtypes = [c_vp, c_int, dp]
    L.cmdr_comp_set_active.argtypes = [c_vp, c_int, c_int]
    L.cmdr_compact_set_active.argtypes = [c_vp, c_int, c_int]
    L.cmdr_cl_update_S.argtypes = [c_int, c_int, c_int, dp, dp, dp, dp, dp]
    L.cmdr_cl_sample_binned.argtypes = [c_int, c_int, dp, dp, dp, c_int, ctypes.POINTER(ClBin), dp, c_int, dp,
                                        ctypes.POINTER(c_int)]
    L.cmdr_cl_sample_lookup.argtypes = [c_int, c_int, c_int, c_int, dp, ctypes.POINTER(c_int), dp, dp, dp, ctypes.c_double, dp,
                                        ctypes.POINTER(c_int)]
    L.cmdr_cl_apod.argtypes = [c_int, c_int, c_int, c_int, c_int]
    L.cmdr_cl_apod.restype = ctypes.c_double
    L.cmdr_cl_apply_apod.argtypes = [c_int, c_int, c_int, c_int, dp, dp, dp]
    L.cmdr_matmulA.argtypes = [c_vp, dp, dp]
    L.cmdr_invM.argtypes = [c_vp, dp, dp]
    L.cmdr_matmulA_dev.argtypes = [c_vp, c_vp, c_vp]
    L.cmdr_invM_dev.argtypes = [c_vp, c_vp, c_vp]
    L.cmdr_compute_rhs.argtypes = [c_vp, c_int, pdp, pdp, dp, dp, dp]
    L.cmdr_compute_rhs_dev.argtypes = [c_vp, c_int, pvp, pvp, c_vp, c_vp, c_vp]
    L.cmdr_compute_residual.argtypes = [c_vp, dp, pdp, pdp]
    L.cmdr_compute_residual_dev.argtypes = [c_vp, c_vp, pvp, pvp]
    L.cmdr_problem_info_ext.argtypes = [c_vp, c_int, ctypes.POINTER(c_i64)]
    L.cmdr_apply_mono_dipole_prior.argtypes = [c_vp, c_int, dp, c_int, dp, dp, c_i64, c_int, dp]
    L.cmdr_apply_mono_dipole_prior_dev.argtypes = [c_vp, c_int, c_vp, c_int, dp, c_vp, c_int, dp]
    pint = ctypes.POINTER(c_int)
    L.cmdr_sigma_l.argtypes = [dp, c_int, c_int, dp]
    L.cmdr_sigma_l_dev.argtypes = [c_vp, c_i64, c_int, c_int, c_vp]
    L.cmdr_profile_enable.argtypes = [c_vp, c_int]
    L.cmdr_profile_read.argtypes = [c_vp, dp, ctypes.POINTER(ctypes.c_longlong)]
    L.cmdr_profile_read_ext.argtypes = [c_vp, c_int, dp, ctypes.POINTER(ctypes.c_longlong)]
    L.cmdr_alm_to_chain_order.argtypes = [dp, c_int, c_int, ctypes.POINTER(ctypes.c_float)]
    L.cmdr_alm_from_chain_order.argtypes = [ctypes.POINTER(ctypes.c_float), c_int, c_int, dp]
    L.cmdr_chain_write_comp.argtypes = [ctypes.c_char_p, c_int, ctypes.c_char_p, dp, c_int, c_int, dp, dp, dp]
    L.cmdr_chain_read_comp.argtypes = [ctypes.c_char_p, c_int, ctypes.c_char_p, c_int, c_int, dp, dp, dp]
    L.cmdr_problem_info.argtypes = [c_vp, ctypes.POINTER(c_i64)]
    L.cmdr_solve.argtypes = [c_vp, dp, dp, c_int, c_dbl, c_int, c_int, c_int, dp, pint, dp, pint]
    L.cmdr_solve_dev.argtypes = [c_vp, c_vp, c_vp, c_int, c_dbl, c_int, c_int, c_int, c_vp, pint, dp, pint]


def lib():
    """Load the library; raise loudly if it was not built (no silent fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise CmdrError("libcmdr_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU path)")
        L = ctypes.CDLL(_SO)
        _sig(L)
        _LIB = L
    return _LIB


def load(path):
    """Load a library exporting the cmdr_* C ABI from an explicit path (tests use this for the host emulation)."""
    L = ctypes.CDLL(path)
    _sig(L)
    return L


def check(rc, L=None):
    if rc < 0:
        raise CmdrError((L or lib()).cmdr_last_error().decode())
    return rc


def device_mem_info():
    """(free, total) bytes of the current device."""
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    check(lib().cmdr_dev_mem_info(ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def device_count():
    return lib().cmdr_device_count()
