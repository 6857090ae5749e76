"""GPU twins of the ``pyamg.amg_core`` relaxation entry points and of SciPy's
``_sparsetools.csr_matvec`` / ``bsr_matvec`` -- same names, same positional arguments
(NumPy host buffers, in-place on ``x`` / ``Yx``), bound through ctypes to Layer 1 of the
C ABI (include/pyamg_amd.h).  This is the stub a maintainer drops in place of
``from pyamg import amg_core`` for the solve path.

Like the reference's pybind11 overload set (relaxation_bind.cpp:708-715, ``.noconvert()``)
every array must already have the right dtype: a mismatch raises ``TypeError``.
"""
from __future__ import annotations

import numpy as np

from . import _capi as capi

__all__ = ["csr_matvec", "bsr_matvec", "gauss_seidel", "sor_gauss_seidel", "bsr_gauss_seidel",
           "jacobi", "bsr_jacobi", "block_jacobi", "block_jacobi_indexed", "block_gauss_seidel", "jacobi_indexed", "gauss_seidel_indexed", "overlapping_schwarz_csr", "gauss_seidel_ne",
           "gauss_seidel_nr", "jacobi_ne", "pinv_array", "standard_aggregation", "fit_candidates"]


def _sfx(Ax, *vals):
    dt = Ax.dtype
    if dt not in (np.float64, np.float32):
        raise TypeError("incompatible function arguments (float32/float64 only on the device path)")
    for v in vals:
        if not isinstance(v, np.ndarray) or v.dtype != dt or not v.flags.c_contiguous:
            raise TypeError("incompatible function arguments (dtype mismatch or non-contiguous array)")
    return "f64" if dt == np.float64 else "f32"


def _idx(*arrs):
    for a in arrs:
        if not isinstance(a, np.ndarray) or a.dtype != np.int32 or not a.flags.c_contiguous:
            raise TypeError("incompatible function arguments (index arrays must be contiguous int32)")


def _csr5(Ap, Aj, Ax, x, b):
    p = capi.ptr
    return (p(Ap), Ap.size, p(Aj), Aj.size, p(Ax), Ax.size, p(x), x.size, p(b), b.size)


def csr_matvec(n_row, n_col, Ap, Aj, Ax, Xx, Yx):
    """Yx += A @ Xx (scipy.sparse._sparsetools.csr_matvec)."""
    _idx(Ap, Aj)
    s = _sfx(Ax, Xx, Yx)
    capi.check(getattr(capi.lib(), f"pamg_csr_matvec_{s}")(int(n_row), int(n_col), capi.ptr(Ap), capi.ptr(Aj),
                                                          capi.ptr(Ax), capi.ptr(Xx), capi.ptr(Yx)), "csr_matvec")


def bsr_matvec(n_brow, n_bcol, R, C, Ap, Aj, Ax, Xx, Yx):
    """Yx += A @ Xx for BSR with R x C blocks (scipy.sparse._sparsetools.bsr_matvec)."""
    _idx(Ap, Aj)
    s = _sfx(Ax, Xx, Yx)
    capi.check(getattr(capi.lib(), f"pamg_bsr_matvec_{s}")(int(n_brow), int(n_bcol), int(R), int(C), capi.ptr(Ap),
                                                          capi.ptr(Aj), capi.ptr(Ax), capi.ptr(Xx), capi.ptr(Yx)),
               "bsr_matvec")


def gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b)
    capi.check(getattr(capi.lib(), f"pamg_gauss_seidel_{s}")(*_csr5(Ap, Aj, Ax, x, b), int(row_start),
                                                            int(row_stop), int(row_step)), "gauss_seidel")


def sor_gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, omega):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b)
    capi.check(getattr(capi.lib(), f"pamg_sor_gauss_seidel_{s}")(*_csr5(Ap, Aj, Ax, x, b), int(row_start),
                                                                int(row_stop), int(row_step), float(omega)),
               "sor_gauss_seidel")


def bsr_gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, blocksize):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b)
    capi.check(getattr(capi.lib(), f"pamg_bsr_gauss_seidel_{s}")(*_csr5(Ap, Aj, Ax, x, b), int(row_start),
                                                                int(row_stop), int(row_step), int(blocksize)),
               "bsr_gauss_seidel")


def jacobi(Ap, Aj, Ax, x, b, temp, row_start, row_stop, row_step, omega):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b, temp, omega)
    capi.check(getattr(capi.lib(), f"pamg_jacobi_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(temp), temp.size,
                                                      int(row_start), int(row_stop), int(row_step),
                                                      capi.ptr(omega), omega.size), "jacobi")


def gauss_seidel_ne(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, Tx, omega):
    """amg_core.gauss_seidel_ne (relaxation.h:875-904)."""
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b, Tx)
    capi.check(getattr(capi.lib(), f"pamg_gauss_seidel_ne_{s}")(*_csr5(Ap, Aj, Ax, x, b), int(row_start), int(row_stop),
                                                               int(row_step), capi.ptr(Tx), Tx.size, float(omega)),
               "gauss_seidel_ne")


def gauss_seidel_nr(Ap, Aj, Ax, x, z, col_start, col_stop, col_step, Tx, omega):
    """amg_core.gauss_seidel_nr (relaxation.h:939-975); Ap/Aj/Ax are the CSC arrays of A, z the residual."""
    _idx(Ap, Aj)
    s = _sfx(Ax, x, z, Tx)
    capi.check(getattr(capi.lib(), f"pamg_gauss_seidel_nr_{s}")(*_csr5(Ap, Aj, Ax, x, z), int(col_start), int(col_stop),
                                                               int(col_step), capi.ptr(Tx), Tx.size, float(omega)),
               "gauss_seidel_nr")


def jacobi_ne(Ap, Aj, Ax, x, b, Tx, temp, row_start, row_stop, row_step, omega):
    """amg_core.jacobi_ne (relaxation.h:811-840); Tx = delta, the row-scaled residual."""
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b, Tx, temp, omega)
    capi.check(getattr(capi.lib(), f"pamg_jacobi_ne_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(Tx), Tx.size, capi.ptr(temp),
                                                         temp.size, int(row_start), int(row_stop), int(row_step),
                                                         capi.ptr(omega), omega.size), "jacobi_ne")


def jacobi_indexed(Ap, Aj, Ax, x, b, indices, omega):
    """amg_core.jacobi_indexed (relaxation.h:382-427)."""
    _idx(Ap, Aj)
    if indices.dtype != np.int32:
        raise TypeError("jacobi_indexed(): incompatible function arguments (indices must be int32)")
    s = _sfx(Ax, x, b, omega)
    capi.check(getattr(capi.lib(), f"pamg_jacobi_indexed_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(indices), indices.size,
                                                              capi.ptr(omega), omega.size), "jacobi_indexed")


def bsr_jacobi(Ap, Aj, Ax, x, b, temp, row_start, row_stop, row_step, blocksize, omega):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b, temp, omega)
    capi.check(getattr(capi.lib(), f"pamg_bsr_jacobi_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(temp), temp.size,
                                                          int(row_start), int(row_stop), int(row_step),
                                                          int(blocksize), capi.ptr(omega), omega.size),
               "bsr_jacobi")


def block_jacobi(Ap, Aj, Ax, x, b, Tx, temp, row_start, row_stop, row_step, omega, blocksize):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b, Tx, temp, omega)
    capi.check(getattr(capi.lib(), f"pamg_block_jacobi_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(Tx), Tx.size,
                                                            capi.ptr(temp), temp.size, int(row_start),
                                                            int(row_stop), int(row_step), capi.ptr(omega),
                                                            omega.size, int(blocksize)), "block_jacobi")


def overlapping_schwarz_csr(Ap, Aj, Ax, x, b, Tx, Tp, Sj, Sp, nsdomains, nrows, row_start, row_stop, row_step):
    """amg_core.overlapping_schwarz_csr (relaxation.h:1420-1492)."""
    _idx(Ap, Aj)
    for a, n in ((Tp, "Tp"), (Sj, "Sj"), (Sp, "Sp")):
        if a.dtype != np.int32:
            raise TypeError(f"overlapping_schwarz_csr(): incompatible function arguments ({n} must be int32)")
    s = _sfx(Ax, x, b, Tx)
    capi.check(getattr(capi.lib(), f"pamg_overlapping_schwarz_csr_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(Tx), Tx.size,
                                                                       capi.ptr(Tp), Tp.size, capi.ptr(Sj), Sj.size,
                                                                       capi.ptr(Sp), Sp.size, int(nsdomains), int(nrows),
                                                                       int(row_start), int(row_stop), int(row_step)),
               "overlapping_schwarz_csr")


def gauss_seidel_indexed(Ap, Aj, Ax, x, b, Id, row_start, row_stop, row_step):
    """amg_core.gauss_seidel_indexed (relaxation.h:736-790)."""
    _idx(Ap, Aj)
    if Id.dtype != np.int32:
        raise TypeError("gauss_seidel_indexed(): incompatible function arguments (Id must be int32)")
    s = _sfx(Ax, x, b)
    capi.check(getattr(capi.lib(), f"pamg_gauss_seidel_indexed_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(Id), Id.size,
                                                                    int(row_start), int(row_stop), int(row_step)),
               "gauss_seidel_indexed")


def block_jacobi_indexed(Ap, Aj, Ax, x, b, Tx, indices, omega, blocksize):
    """amg_core.block_jacobi_indexed (relaxation.h:1129-1199)."""
    _idx(Ap, Aj)
    if indices.dtype != np.int32:
        raise TypeError("block_jacobi_indexed(): incompatible function arguments (indices must be int32)")
    s = _sfx(Ax, x, b, Tx, omega)
    capi.check(getattr(capi.lib(), f"pamg_block_jacobi_indexed_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(Tx), Tx.size,
                                                                    capi.ptr(indices), indices.size, capi.ptr(omega),
                                                                    omega.size, int(blocksize)), "block_jacobi_indexed")


def block_gauss_seidel(Ap, Aj, Ax, x, b, Tx, row_start, row_stop, row_step, blocksize):
    _idx(Ap, Aj)
    s = _sfx(Ax, x, b, Tx)
    capi.check(getattr(capi.lib(), f"pamg_block_gauss_seidel_{s}")(*_csr5(Ap, Aj, Ax, x, b), capi.ptr(Tx), Tx.size,
                                                                  int(row_start), int(row_stop), int(row_step),
                                                                  int(blocksize)), "block_gauss_seidel")


def pinv_array(AA, m, n, TransA):
    """amg_core.pinv_array (linalg.h:930-1000): the (m, n, n) array AA, passed ravelled like the reference's callers do
    (util/utils.py:684), is overwritten block by block with the pseudo-inverses.  n <= 6."""
    if not isinstance(AA, np.ndarray) or AA.dtype not in (np.float64, np.float32) or not AA.flags.c_contiguous:
        raise TypeError("incompatible function arguments (contiguous float32/float64 array expected)")
    s = "f64" if AA.dtype == np.float64 else "f32"
    t = TransA if isinstance(TransA, bytes) else str(TransA).encode()
    capi.check(getattr(capi.lib(), f"pamg_pinv_array_{s}")(capi.ptr(AA), AA.size, int(m), int(n), t[:1]), "pinv_array")


def standard_aggregation(n_row, Ap, Aj, x, y):
    """amg_core.standard_aggregation (smoothed_aggregation.h:137-268): x <- aggregate of every node (-1: none), y <- root
    nodes; returns the number of aggregates, like the reference."""
    import ctypes
    _idx(Ap, Aj, x, y)
    na = ctypes.c_int(0)
    capi.check(capi.lib().pamg_standard_aggregation(int(n_row), capi.ptr(Ap), Ap.size, capi.ptr(Aj), Aj.size, capi.ptr(x), x.size,
                                                   capi.ptr(y), y.size, ctypes.byref(na)), "standard_aggregation")
    return int(na.value)


def fit_candidates(n_row, n_col, K1, K2, Ap, Ai, Ax, B, R, tol):
    """amg_core.fit_candidates (real; smoothed_aggregation.h:484-660): Ap / Ai = CSC arrays of AggOp, Ax (ravelled
    (nnz, K1, K2)) and R (ravelled (n_col, K2, K2)) are overwritten."""
    _idx(Ap, Ai)
    s = _sfx(Ax, B, R)
    capi.check(getattr(capi.lib(), f"pamg_fit_candidates_{s}")(int(n_row), int(n_col), int(K1), int(K2), capi.ptr(Ap), Ap.size,
                                                              capi.ptr(Ai), Ai.size, capi.ptr(Ax), Ax.size, capi.ptr(B), B.size,
                                                              capi.ptr(R), R.size, float(tol)), "fit_candidates")
