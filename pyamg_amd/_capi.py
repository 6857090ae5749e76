"""ctypes binding of libpyamg_amd.so (the C ABI declared in include/pyamg_amd.h).

The product path has no CPU fallback: if the HIP library is missing or no MI355X is
visible, every entry point raises (``DeviceUnavailable``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("PAMG_LIB", PKG / "libpyamg_amd.so"))    # PAMG_LIB: experiment builds only

OK = 0
E_ARG, E_UNSUPPORTED, E_NODEVICE, E_STATE, E_ALLOC = -1, -2, -3, -4, -5
E_TIMEOUT, E_COMM = -6, -7
F64, F32 = 0, 1
CSR, BSR = 0, 1
SPMV_SET, SPMV_ACC, SPMV_RESID, SPMV_AXPBY, SPMV_ACC_AXPBY = 0, 1, 2, 3, 4
FORWARD, BACKWARD, SYMMETRIC = 0, 1, 2
SMOOTH = {"schwarz": 14, "cf_block_jacobi": 12, "fc_block_jacobi": 13, "gauss_seidel_ne": 9, "gauss_seidel_nr": 10, "jacobi_ne": 11, "cf_jacobi": 7, "fc_jacobi": 8, "none": 0, "jacobi": 1, "gauss_seidel": 2, "sor": 3, "polynomial": 4,
          "block_jacobi": 5, "block_gauss_seidel": 6}
KRYLOV = {"cg": 0, "gmres": 1, "cgne": 2, "cgnr": 3}
SWEEP = {"forward": FORWARD, "backward": BACKWARD, "symmetric": SYMMETRIC}
CYCLE = {"V": 0, "W": 1, "F": 2, "AMLI": 3}


class DeviceUnavailable(RuntimeError):
    """libpyamg_amd.so is not built/loadable or no HIP device is visible."""


class PamgError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        super().__init__(f"{where}: status {status} ({status_string(status)})")


_lib = None
_vp = C.c_void_p
_i = C.c_int
_d = C.c_double
_sz = C.c_size_t
_i32p = C.c_void_p


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float64:
        return F64
    if dt == np.float32:
        return F32
    raise TypeError(f"dtype {dt} is not supported on the device path (float64/float32)")


def _declare(lib):
    def f(name, *args, res=_i):
        fn = getattr(lib, name)
        fn.argtypes = list(args)
        fn.restype = res
        return fn
    P = C.POINTER
    f("pamg_version", res=C.c_char_p)
    f("pamg_status_string", _i, res=C.c_char_p)
    f("pamg_device_count", P(_i))
    f("pamg_set_device", _i)
    f("pamg_get_device", P(_i))
    f("pamg_bandwidth_probe", _i, C.c_int64, _i, P(C.c_double))
    f("pamg_device_name", _i, C.c_char_p, _i)
    f("pamg_malloc", P(_vp), _sz)
    f("pamg_free", _vp)
    for n in ("h2d", "d2h", "d2d"):
        f(f"pamg_memcpy_{n}", _vp, _vp, _sz, _vp)
    f("pamg_memset", _vp, _i, _sz, _vp)
    f("pamg_stream_create", P(_vp))
    f("pamg_stream_destroy", _vp)
    f("pamg_stream_synchronize", _vp)
    f("pamg_device_synchronize")
    f("pamg_event_create", P(_vp))
    f("pamg_event_destroy", _vp)
    f("pamg_event_record", _vp, _vp)
    f("pamg_event_synchronize", _vp)
    f("pamg_event_elapsed_ms", _vp, _vp, P(C.c_float))
    for sfx, ct in (("f64", C.c_double), ("f32", C.c_float)):
        f(f"pamg_csr_matvec_{sfx}", _i, _i, _vp, _vp, _vp, _vp, _vp)
        f(f"pamg_bsr_matvec_{sfx}", _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp)
        csr5 = (_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i)
        f(f"pamg_gauss_seidel_{sfx}", *csr5, _i, _i, _i)
        f(f"pamg_sor_gauss_seidel_{sfx}", *csr5, _i, _i, _i, ct)
        f(f"pamg_overlapping_schwarz_csr_{sfx}", *csr5, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i)
        f(f"pamg_gauss_seidel_indexed_{sfx}", *csr5, _vp, _i, _i, _i, _i)
        f(f"pamg_bsr_gauss_seidel_{sfx}", *csr5, _i, _i, _i, _i)
        f(f"pamg_jacobi_{sfx}", *csr5, _vp, _i, _i, _i, _i, _vp, _i)
        f(f"pamg_bsr_jacobi_{sfx}", *csr5, _vp, _i, _i, _i, _i, _i, _vp, _i)
        f(f"pamg_jacobi_indexed_{sfx}", *csr5, _vp, _i, _vp, _i)
        f(f"pamg_gauss_seidel_ne_{sfx}", *csr5, _i, _i, _i, _vp, _i, ct)
        f(f"pamg_gauss_seidel_nr_{sfx}", *csr5, _i, _i, _i, _vp, _i, ct)
        f(f"pamg_jacobi_ne_{sfx}", *csr5, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i)
        f(f"pamg_block_jacobi_{sfx}", *csr5, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _i)
        f(f"pamg_block_gauss_seidel_{sfx}", *csr5, _vp, _i, _i, _i, _i, _i)
        f(f"pamg_block_jacobi_indexed_{sfx}", *csr5, _vp, _i, _vp, _i, _vp, _i, _i)
    f("pamg_pinv_array_f64", _vp, _i, _i, _i, C.c_char)
    f("pamg_pinv_array_f32", _vp, _i, _i, _i, C.c_char)
    f("pamg_dev_pinv_array", _i, _vp, C.c_int64, _i, _i, _vp)
    f("pamg_bsr_transpose_f64", _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp)
    f("pamg_bsr_transpose_f32", _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp)
    f("pamg_standard_aggregation", _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, P(_i))
    f("pamg_csr_standard_aggregation", _vp, _vp, _vp, P(_i))
    f("pamg_fit_candidates_f64", _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _d)
    f("pamg_fit_candidates_f32", _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, C.c_float)
    f("pamg_fit_tentative_f64", _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _d)
    f("pamg_fit_tentative_f32", _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_float)
    f("pamg_matrix_create", P(_vp), _i, _i, _i, _i, _i, _i, _vp, _vp, _vp)
    f("pamg_matrix_destroy", _vp)
    f("pamg_matrix_info", _vp, P(C.c_int64))
    f("pamg_matrix_tune", _vp, _i, _i)
    f("pamg_matrix_value_codes", _vp, P(C.c_int))
    f("pamg_matrix_row_patterns", _vp, P(C.c_int))
    f("pamg_matrix_row_masks", _vp, P(C.c_longlong))
    f("pamg_matrix_flow_error", _vp, P(_i))
    f("pamg_matrix_subset_rows", _vp, _vp, _i, P(_vp))
    f("pamg_matrix_kaczmarz", _vp, _i, _vp, _vp, _vp, _d, _i, _i, _vp, _vp)
    f("pamg_vec_mul", _i, C.c_int64, _vp, _vp, _vp, _vp)
    f("pamg_matrix_jacobi_indexed", _vp, _vp, _vp, _d, _vp, _vp)
    f("pamg_matrix_gs_profile", _vp, _i, _vp, C.c_int64, P(C.c_int64))
    f("pamg_matrix_tile_info", _vp, _i, P(C.c_int64))
    f("pamg_matrix_lane_info", _vp, _i, P(C.c_int64))
    f("pamg_matrix_lanem_info", _vp, _i, P(C.c_int64), P(C.c_double))
    f("pamg_matrix_point_twin", _vp, P(_i))
    f("pamg_matrix_lanem_levels", _vp, _i, _vp, C.c_int64, P(C.c_int64))
    f("pamg_matrix_kz_info", _vp, _i, P(C.c_int64))
    f("pamg_matrix_line_info", _vp, _i, P(C.c_int64))
    f("pamg_matrix_lane_profile", _vp, _i, _vp, C.c_int64, P(C.c_int64))
    f("pamg_matrix_autotune", _vp, _i)
    f("pamg_matrix_spmv", _vp, _i, _vp, _vp, _d, _vp, _vp)
    f("pamg_matrix_split_ranges", _vp, C.c_int64)
    f("pamg_matrix_spmv_part", _vp, _i, _i, _vp, _vp, _d, _vp, _vp)
    f("pamg_matrix_resid_sumsq", _vp, _vp, _vp, _vp, _vp)
    f("pamg_matrix_jacobi", _vp, _vp, _vp, _vp, _d, _i, _vp)
    f("pamg_l1_cache_clear")
    f("pamg_l1_cache_size", P(C.c_int))
    f("pamg_matrix_jacobi_step", _vp, _vp, _vp, _vp, _d, _vp)
    f("pamg_matrix_block_jacobi_step", _vp, _vp, _vp, _vp, _vp, _d, _vp)
    f("pamg_matrix_block_jacobi_indexed", _vp, _vp, _vp, _vp, _vp, C.c_int64, _d, _vp, _vp)
    f("pamg_schwarz_create", P(_vp), _vp, _i, _vp, _vp, _vp, _vp)
    f("pamg_schwarz_destroy", _vp)
    f("pamg_schwarz_sweep", _vp, _vp, _vp, _i, _i, _i, _vp)
    f("pamg_schwarz_info", _vp, P(C.c_int64))
    f("pamg_schwarz_set_mode", _vp, _i)
    f("pamg_schwarz_error", _vp, P(_i))
    f("pamg_solver_set_schwarz_smoother", _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp)
    f("pamg_solver_set_cf_block_smoother", _vp, _i, _i, _i, _i, _i, _i, _d, _vp, _i, _vp, _i, _vp, _i)
    f("pamg_matrix_gauss_seidel", _vp, _vp, _vp, _i, _d, _i, _vp)
    f("pamg_matrix_polynomial", _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp)
    f("pamg_matrix_block_jacobi", _vp, _vp, _vp, _vp, _vp, _d, _i, _vp)
    f("pamg_matrix_block_gauss_seidel", _vp, _vp, _vp, _vp, _i, _i, _vp)
    f("pamg_vec_sumsq", _i, C.c_int64, _vp, _vp, _vp)
    f("pamg_vec_axpy", _i, C.c_int64, _d, _vp, _vp, _vp)
    f("pamg_vec_scale", _i, C.c_int64, _d, _vp, _vp, _vp)
    f("pamg_vec_gather", _i, C.c_int64, _vp, _vp, _vp, _vp)
    f("pamg_csr_renumber", _i, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp)
    f("pamg_csr_row_argmax_abs", _i, C.c_int64, _vp, _vp, _vp, _vp)
    f("pamg_csr_sort_rows", _i, C.c_int64, _vp, _vp, _vp, _i)
    f("pamg_host_cpus", _i)
    f("pamg_solver_create", P(_vp), _i)
    f("pamg_solver_destroy", _vp)
    f("pamg_solver_add_level", _vp, _vp, _vp, _vp)
    f("pamg_solver_set_smoother", _vp, _i, _i, _i, _i, _d, _i, _vp, _i, _vp, _i)
    f("pamg_solver_set_ne_smoother", _vp, _i, _i, _i, _i, _d, _i, _vp, _vp, _vp)
    f("pamg_solver_set_krylov_smoother", _vp, _i, _i, _i, _d, _i, _i, _vp)
    f("pamg_solver_set_cf_smoother", _vp, _i, _i, _i, _i, _i, _i, _d, _vp, _i, _vp, _i)
    f("pamg_solver_set_coarse_dense", _vp, _vp, _i)
    f("pamg_solver_set_coarse_relax", _vp)
    f("pamg_solver_set_coarse_host", _vp, _vp, _vp, _i)
    f("pamg_solver_finalize", _vp)
    f("pamg_solver_cycle", _vp, _vp, _vp, _i, _i, _vp)
    f("pamg_solver_solve", _vp, _vp, _vp, _d, _i, _i, _i, _i, _vp, P(_i), P(_i), _vp)
    f("pamg_solver_set_graph", _vp, _i)
    f("pamg_solver_pcg", _vp, _vp, _vp, _d, _i, _i, _i, _vp, P(_i), P(_i), _vp)
    f("pamg_solver_fgmres", _vp, _vp, _vp, _d, _i, _i, _i, _i, _vp, _i, P(_i), P(_i), P(_i), _vp)
    f("pamg_solver_gmres", _vp, _vp, _vp, _d, _i, _i, _i, _i, _vp, _i, P(_i), P(_i), P(_i), _vp)
    f("pamg_solver_load", _vp, _vp, _vp, _vp)
    f("pamg_solver_iterate", _vp, _i, _i, _i, _vp, _vp)
    f("pamg_solver_store", _vp, _vp, _vp)
    f("pamg_solver_stream", _vp, P(_vp))
    f("pamg_solver_stats", _vp, P(C.c_int64))
    f("pamg_dist_create", P(_vp), _i, _i, _i)
    f("pamg_dist_destroy", _vp)
    f("pamg_dist_add_level", _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, _i, _vp, _vp, _vp, _i, _vp, _vp)
    f("pamg_dist_set_collapse", _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp)
    f("pamg_dist_set_smoother", _vp, _i, _i, _i, _i, _d, _vp, _i, _vp, _i)
    f("pamg_dist_set_callbacks", _vp, _vp, _vp, _vp)
    f("pamg_dist_rccl_unique_id", _vp)
    f("pamg_dist_set_rccl", _vp, _vp)
    f("pamg_dist_finalize", _vp)
    f("pamg_dist_set_options", _vp, _i, _i)
    f("pamg_dist_load", _vp, _vp, _vp)
    f("pamg_dist_store", _vp, _vp)
    f("pamg_dist_iterate", _vp, _i, _vp)
    f("pamg_dist_resid_norm", _vp, P(_d))
    f("pamg_dist_sync", _vp)
    f("pamg_dist_stream", _vp, P(_vp))
    f("pamg_dist_info", _vp, P(C.c_int64))
    f("pamg_dist_set_allgather", _vp, _i, C.c_int64, _vp)
    f("pamg_dist_set_exchange", _vp, _i)
    f("pamg_dist_set_model_transport", _vp)
    f("pamg_dist_level_info", _vp, _i, P(C.c_int64))
    f("pamg_rccl_selftest", C.c_int64, P(C.c_double))
    f("pamg_rccl_available")
    f("pamg_dist_exchange_test", _vp, _i, _vp, _vp)
    f("pamg_csr_create", P(_vp), C.c_int64, C.c_int64, _vp, _vp, _vp)
    f("pamg_csr_view", P(_vp), _vp)
    f("pamg_csr_destroy", _vp)
    f("pamg_csr_info", _vp, P(C.c_int64))
    f("pamg_csr_download", _vp, _vp, _vp, _vp)
    f("pamg_csr_matmat", _vp, _vp, _i, _i, P(_vp))
    f("pamg_csr_subtract", _vp, _vp, P(_vp))
    f("pamg_csr_subtract_bsr", _vp, _vp, _i, _i, P(_vp))
    f("pamg_csr_scale", _vp, _d)
    f("pamg_csr_strength_symmetric", _vp, _d, P(_vp))
    f("pamg_matrix_scale_rows", _vp, _vp)
    f("pamg_matrix_scale_values", _vp, _d)
    f("pamg_arnoldi_create", P(_vp), _vp, _i)
    f("pamg_arnoldi_destroy", _vp)
    f("pamg_arnoldi_run", _vp, _vp, _vp, _d, _vp, P(_i), P(_i))
    f("pamg_arnoldi_combine", _vp, _i, _vp, _vp)
    f("pamg_arnoldi_vector", _vp, _vp, _vp, P(_i))


def load(require_device: bool = False):
    """Load the shared library (no compute).  ``require_device`` additionally demands a
    visible HIP device."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise DeviceUnavailable(
                f"{LIB_PATH} not found: build it with `python -m pyamg_amd._build` "
                "(the MI355X engine has no CPU fallback)")
        try:
            lib = C.CDLL(str(LIB_PATH))
        except OSError as e:          # pragma: no cover
            raise DeviceUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        _declare(lib)
        _lib = lib
    if require_device and device_count() < 1:
        raise DeviceUnavailable("no HIP device visible: the MI355X engine has no CPU fallback")
    return _lib


def lib():
    return load(require_device=True)


def status_string(st: int) -> str:
    try:
        return load().pamg_status_string(int(st)).decode()
    except Exception:       # pragma: no cover
        return "?"


def check(st: int, where: str = ""):
    if st != OK:
        if st == E_UNSUPPORTED:
            raise NotImplementedError(f"{where}: not supported on the device path")
        raise PamgError(st, where)


def device_count() -> int:
    n = _i(0)
    st = load().pamg_device_count(C.byref(n))
    return int(n.value) if st == OK else 0


def version() -> str:
    return load().pamg_version().decode()


def ptr(a) -> C.c_void_p:
    """Host pointer of a NumPy array (None -> NULL)."""
    if a is None:
        return C.c_void_p(0)
    return C.c_void_p(a.ctypes.data)


# ------------------------------------------------------------------ device memory helper
class DeviceArray:
    """A typed 1-D device buffer owned through pamg_malloc/pamg_free."""

    def __init__(self, n: int, dtype):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        p = _vp()
        check(lib().pamg_malloc(C.byref(p), max(self.n, 1) * self.dtype.itemsize), "pamg_malloc")
        self.ptr = p

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype).reshape(-1)
        assert a.size == self.n
        check(lib().pamg_memcpy_h2d(self.ptr, ptr(a), a.nbytes, None), "h2d")

    def download(self) -> np.ndarray:
        out = np.empty(self.n, dtype=self.dtype)
        check(lib().pamg_memcpy_d2h(ptr(out), self.ptr, out.nbytes, None), "d2h")
        return out

    def zero(self):
        check(lib().pamg_memset(self.ptr, 0, self.n * self.dtype.itemsize, None), "memset")
        check(lib().pamg_device_synchronize(), "sync")

    def free(self):
        if getattr(self, "ptr", None) is not None and self.ptr:
            try:
                _lib.pamg_free(self.ptr)
            except Exception:   # pragma: no cover
                pass
            self.ptr = None

    def __del__(self):
        self.free()


def bandwidth_probe(kind: str = "copy", n: int = 1 << 27, reps: int = 20) -> float:
    """measured GB/s of an on-device copy / triad with 16-byte accesses (pamg_bandwidth_probe)"""
    out = C.c_double(0.0)
    kinds = {"copy": 0, "triad": 1, "copy1": 2, "copy4": 3, "copy4nt": 4, "copy8": 5, "read": 6, "write": 7, "memcpy": 8, "copy8b": 9, "copy8bnt": 10, "copy1nt": 11, "read_one_xcd": 12, "read_one_xcd_2wg": 13}
    check(lib().pamg_bandwidth_probe(kinds[kind], int(n), int(reps), C.byref(out)), "pamg_bandwidth_probe")
    return float(out.value)


def sync():
    check(lib().pamg_device_synchronize(), "sync")


class Event:
    def __init__(self):
        self.e = _vp()
        check(lib().pamg_event_create(C.byref(self.e)), "event_create")

    def record(self, stream=None):
        check(lib().pamg_event_record(self.e, stream), "event_record")

    def synchronize(self):
        check(lib().pamg_event_synchronize(self.e), "event_sync")

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(lib().pamg_event_elapsed_ms(self.e, stop.e, C.byref(ms)), "event_elapsed")
        return float(ms.value)

    def __del__(self):
        try:
            _lib.pamg_event_destroy(self.e)
        except Exception:       # pragma: no cover
            pass
