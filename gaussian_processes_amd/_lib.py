"""ctypes binding of libgpx.so (include/gpx.h) -- the only door to the GPU.

There is no CPU fallback: if the shared library is missing or no MI355X is
usable, every compute call raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C gaussian_processes_amd/csrc``.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t,
                    c_uint64, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpx.so")

OK = 0
ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_NOMEM, ERR_UNSUPPORTED, ERR_INTERNAL = -1, -2, -3, -4, -5, -6
(ROUTE_TRSV_OPS, ROUTE_TRSV_STEPS, ROUTE_PANEL_RES, ROUTE_PANEL_CHAIN, ROUTE_FIT_RIDE, ROUTE_FIT_TWO_SOLVES,
 ROUTE_GEMM_FAST, ROUTE_GEMM_GENERIC, ROUTE_SYRK_EXACT, ROUTE_SYRK_PATCH, ROUTE_MG_BCAST_ONE, ROUTE_MG_BCAST_SAG,
 ROUTE_FIT_OPS_AHEAD, ROUTE_TRSM_OPS, ROUTE_POTRF_PAIR) = range(15)
F64, F32 = 0, 1
KERNEL_GAUSSIAN, KERNEL_PERIODIC = 0, 1
FULL, LOWER = 0, 1
(K, DK_DH, DK_DW, DK_DP, D2K_DHDH, D2K_DHDW, D2K_DHDP, D2K_DWDW, D2K_DWDP, D2K_DPDP) = range(10)
MIN_LOG = -705.6238298100243

MEMBER_BY_NAME = {
    "K": K, "dK_dh": DK_DH, "dK_dw": DK_DW, "dK_dp": DK_DP,
    "d2K_dhdh": D2K_DHDH, "d2K_dhdw": D2K_DHDW, "d2K_dwdh": D2K_DHDW,
    "d2K_dhdp": D2K_DHDP, "d2K_dpdh": D2K_DHDP, "d2K_dwdw": D2K_DWDW,
    "d2K_dwdp": D2K_DWDP, "d2K_dpdw": D2K_DWDP, "d2K_dpdp": D2K_DPDP,
}

MG_ID_BYTES = 128
MG_BCAST_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_void_p)
MG_ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p)

c_double_p = POINTER(c_double)
c_int_p = POINTER(c_int)

# name -> (restype, argtypes); mirrors include/gpx.h one to one
_SIGNATURES = {
    "gpx_version": (c_int, []),
    "gpx_last_error": (c_char_p, []),
    "gpx_device_count": (c_int, [c_int_p]),
    "gpx_set_device": (c_int, [c_int]),
    "gpx_get_device": (c_int, [c_int_p]),
    "gpx_device_info": (c_int, [c_int, c_char_p, c_size_t, c_int_p, c_int_p, POINTER(c_uint64)]),
    "gpx_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "gpx_free": (c_int, [c_void_p]),
    "gpx_mem_info": (c_int, [ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "gpx_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpx_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpx_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpx_memcpy2d_h2d": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "gpx_memcpy2d_d2h": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p]),
    "gpx_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "gpx_stream_create": (c_int, [POINTER(c_void_p)]),
    "gpx_stream_destroy": (c_int, [c_void_p]),
    "gpx_stream_sync": (c_int, [c_void_p]),
    "gpx_device_sync": (c_int, []),
    "gpx_event_create": (c_int, [POINTER(c_void_p)]),
    "gpx_event_destroy": (c_int, [c_void_p]),
    "gpx_event_record": (c_int, [c_void_p, c_void_p]),
    "gpx_event_sync": (c_int, [c_void_p]),
    "gpx_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "gpx_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "gpx_prof_enable": (c_int, [c_int]),
    "gpx_prof_read": (c_int, [c_int, c_double_p, c_double_p, c_double_p]),
    "gpx_d_kmat": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int,
                           c_double_p, c_double, c_int, c_void_p, c_int64, c_void_p]),
    "gpx_d_mean": (c_int, [c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_double_p,
                           c_void_p, c_void_p, c_void_p]),
    "gpx_d_mean_member": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_double_p,
                                  c_void_p, c_void_p, c_void_p]),
    "gpx_d_gemm_nt": (c_int, [c_int, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64,
                              c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64,
                              c_void_p]),
    "gpx_d_potrf": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpx_d_potrf_panel": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                  c_void_p]),
    "gpx_d_syrk_bc": (c_int, [c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                              c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gpx_d_tril": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p]),
    "gpx_d_trsv_lower": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                 c_void_p]),
    "gpx_d_trsv_lower_cols": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                      c_void_p]),
    "gpx_d_panel_gemv_t": (c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    "gpx_d_trsm_right_lt": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                    c_void_p]),
    "gpx_d_logdet_chol": (c_int, [c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpx_d_dot": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "gpx_gp_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int64, c_int]),
    "gpx_gp_destroy": (c_int, [c_void_p]),
    "gpx_gp_set_data": (c_int, [c_void_p, c_double_p, c_double_p]),
    "gpx_gp_set_data_device": (c_int, [c_void_p, c_void_p, c_void_p]),
    "gpx_gp_set_params": (c_int, [c_void_p, c_double_p, c_double]),
    "gpx_gp_set_K": (c_int, [c_void_p, c_double_p, c_int64]),
    "gpx_gp_fit": (c_int, [c_void_p, c_int_p]),
    "gpx_gp_log_lh": (c_int, [c_void_p, c_double_p]),
    "gpx_gp_logdet": (c_int, [c_void_p, c_double_p]),
    "gpx_gp_info": (c_int, [c_void_p, c_int_p]),
    "gpx_gp_mean": (c_int, [c_void_p, c_double_p, c_int64, c_double_p]),
    "gpx_gp_cov": (c_int, [c_void_p, c_double_p, c_int64, c_double_p]),
    "gpx_gp_mean_from_K": (c_int, [c_void_p, c_double_p, c_int64, c_double_p]),
    "gpx_gp_cov_from_K": (c_int, [c_void_p, c_double_p, c_double_p, c_int64, c_double_p]),
    "gpx_gp_get_Kxx": (c_int, [c_void_p, c_double_p, c_int64]),
    "gpx_gp_get_Lxx": (c_int, [c_void_p, c_double_p, c_int64]),
    "gpx_gp_get_alpha": (c_int, [c_void_p, c_double_p]),
    "gpx_gp_get_inv_Kxx": (c_int, [c_void_p, c_double_p, c_int64]),
    "gpx_gp_dloglh_dtheta": (c_int, [c_void_p, c_double_p]),
    "gpx_gp_dlh_d2lh": (c_int, [c_void_p, c_double_p, c_double_p, c_double_p]),
    "gpx_gp_dm_dtheta": (c_int, [c_void_p, c_double_p, c_int64, c_double_p]),
    "gpx_gp_fit_batch": (c_int, [c_void_p, c_double_p, c_int64, c_double_p, c_int_p]),
    "gpx_gp_fit_batch_grad": (c_int, [c_void_p, c_double_p, c_int64, c_double_p, c_double_p, c_double_p, c_int_p]),
    "gpx_gp_save": (c_int, [c_void_p, c_char_p]),
    "gpx_gp_load": (c_int, [POINTER(c_void_p), c_char_p]),
    "gpx_gp_describe": (c_int, [c_void_p, c_int_p, c_int_p, POINTER(c_int64), c_int_p, c_double_p, c_double_p]),
    "gpx_gp_get_xy": (c_int, [c_void_p, c_double_p, c_double_p]),
    "gpx_gp_last_timing": (c_int, [c_void_p, POINTER(c_float)]),
    "gpx_gp_device_ptrs": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                   POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "gpx_debug_route_count": (c_int, [c_int, POINTER(c_int64)]),
    "gpx_debug_roctx_ranges": (c_int, [POINTER(c_int64)]),
    "gpx_debug_tune_refreshes": (c_int, [POINTER(c_int64)]),
    "gpx_debug_leaf_selfcheck": (c_int, [c_int, POINTER(c_int)]),
    "gpx_debug_route_reset": (c_int, []),
    "gpx_debug_mg_inject_info": (c_int, [c_void_p, c_int]),
    "gpx_debug_mg_plan": (c_int, [c_int64, c_int64, c_int, c_int, c_int64, POINTER(c_int64), c_int]),
    "gpx_mg_probe": (c_int, []),
    "gpx_mg_create_local": (c_int, [POINTER(c_void_p), c_int, c_int, c_int64, c_int, c_int64, c_int, c_int]),
    "gpx_mg_connect": (c_int, [c_void_p, c_void_p]),
    "gpx_mg_comm_info": (c_int, [c_void_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "gpx_mg_set_bcast": (c_int, [c_void_p, c_int]),
    "gpx_mg_unique_id": (c_int, [c_void_p]),
    "gpx_mg_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int64, c_int, c_int64, c_int, c_int, c_void_p]),
    "gpx_mg_create_cb": (c_int, [POINTER(c_void_p), c_int, c_int, c_int64, c_int, c_int64, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "gpx_mg_destroy": (c_int, [c_void_p]),
    "gpx_mg_set_data": (c_int, [c_void_p, c_double_p, c_double_p]),
    "gpx_mg_fit": (c_int, [c_void_p, c_double_p, c_double, c_double_p, c_int_p]),
    "gpx_mg_mean": (c_int, [c_void_p, c_double_p, c_double_p, c_int64, c_double_p]),
    "gpx_mg_get_alpha": (c_int, [c_void_p, c_double_p]),
    "gpx_mg_scalars": (c_int, [c_void_p, c_double_p, c_double_p, c_int_p]),
    "gpx_mg_timing": (c_int, [c_void_p, c_double_p]),
    "gpx_mg_timing_ex": (c_int, [c_void_p, c_double_p, c_int]),
    "gpx_mg_chain_by_panel": (c_int, [c_void_p, c_double_p, c_int64]),
    "gpx_mg_adopt_comm": (c_int, [c_void_p, c_void_p]),
    "gpx_mg_set_chunks": (c_int, [c_void_p, c_int]),
    "gpx_mg_set_owner_first": (c_int, [c_void_p, c_int]),
    "gpx_mg_set_wait_timing": (c_int, [c_void_p, c_int]),
    "gpx_mg_schedule_info": (c_int, [c_void_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "gpx_mg_device_ptrs": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "gpx_mg_create_rehearsal": (c_int, [POINTER(c_void_p), c_int, c_int, c_int64, c_int, c_int64, c_int, c_int, c_void_p, c_int64,
                                        c_void_p, c_double, c_double]),
    "gpx_gaussian_c": (c_int, [c_int, c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                               c_double, c_double]),
    "gpx_gaussian_c_jacobian": (c_int, [c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                                        c_double, c_double]),
    "gpx_gaussian_c_hessian": (c_int, [c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                                       c_double, c_double]),
    "gpx_periodic_c": (c_int, [c_int, c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                               c_double, c_double, c_double]),
    "gpx_periodic_c_jacobian": (c_int, [c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                                        c_double, c_double, c_double]),
    "gpx_periodic_c_hessian": (c_int, [c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                                       c_double, c_double, c_double]),
    "gpx_kmat_host": (c_int, [c_int, c_int, c_double_p, c_double_p, c_int64, c_double_p, c_int64,
                              c_int, c_double_p, c_double]),
    "gpx_cholesky": (c_int, [c_double_p, c_double_p, c_int64, c_int_p]),
    "gpx_cho_solve": (c_int, [c_double_p, c_int64, c_double_p]),
    "gpx_gp_c_log_lh": (c_int, [c_double_p, c_double_p, c_double_p, c_int64, c_double_p]),
    "gpx_gp_c_dloglh_dtheta": (c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double, c_int64, c_int,
                                       c_double_p]),
    "gpx_gp_c_dlh_dtheta": (c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double, c_double, c_int64, c_int,
                                    c_double_p]),
    "gpx_gp_c_d2lh_dtheta2": (c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double, c_double,
                                      c_double_p, c_int64, c_int, c_double_p]),
    "gpx_gp_c_dm_dtheta": (c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double, c_int64, c_int,
                                   c_int64, c_double_p]),
    "gpx_gemm_nt_host": (c_int, [c_double_p, c_double_p, c_double_p, c_int64, c_int64, c_int64]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


class GpxError(RuntimeError):
    pass


def load():
    """Load libgpx.so (once).  Fails loudly when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpxError(
            "libgpx.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C gaussian_processes_amd/csrc` -- there is no CPU fallback"
            % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().gpx_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a gpx status code onto the exception type the reference's callers expect."""
    if rc == OK:
        return
    msg = last_error() or ("gpx status %d" % rc)
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NO_DEVICE:
        raise GpxError("no usable MI355X: " + msg)
    raise GpxError(msg)


def dptr(a):
    """double* view of a C-contiguous float64 ndarray."""
    return a.ctypes.data_as(c_double_p)


def device_count():
    n = c_int(0)
    check(load().gpx_device_count(ctypes.byref(n)))
    return n.value


def device_info(device=0):
    name = ctypes.create_string_buffer(256)
    cus, mhz, mem = c_int(0), c_int(0), c_uint64(0)
    check(load().gpx_device_info(device, name, 256, ctypes.byref(cus), ctypes.byref(mhz),
                                 ctypes.byref(mem)))
    return {"name": name.value.decode(), "cus": cus.value, "clock_mhz": mhz.value,
            "hbm_bytes": mem.value}


def mem_free():
    """Free HBM on the current device, bytes."""
    f, t = c_size_t(0), c_size_t(0)
    check(load().gpx_mem_info(ctypes.byref(f), ctypes.byref(t)))
    return int(f.value)


def route_count(route):
    """How often host-side route `route` (ROUTE_*) was taken since the last `route_reset()`."""
    v = c_int64(0)
    check(load().gpx_debug_route_count(int(route), ctypes.byref(v)))
    return v.value


def route_reset():
    check(load().gpx_debug_route_reset())


def lapack_info_error(info):
    """The LinAlgError scipy.linalg.cholesky raises for a LAPACK info > 0.  (A host-side info is never
    negative: an internal failure of the factorisation arrives as the status GPX_ERR_INTERNAL -> GpxError.)"""
    if info < 0:
        raise GpxError("internal failure inside the factorisation (info = %d)" % info)
    return np.linalg.LinAlgError(
        "%d-th leading minor of the array is not positive definite" % info)
