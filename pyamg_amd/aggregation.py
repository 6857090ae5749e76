"""Setup-phase operators on the device: drop-ins for the three places the reference's
smoothed-aggregation setup spends its time in.

* ``approximate_spectral_radius``   -- pyamg/util/linalg.py:255-370 (Arnoldi of :154-253 on the device)
* ``jacobi_prolongation_smoother``  -- pyamg/aggregation/smooth.py:61-207 (weighting 'diagonal' / 'local')
* ``richardson_prolongation_smoother`` -- smooth.py:210-272
* ``galerkin_product(R, A, P)``     -- the ``R @ A @ P`` of aggregation.py:425 / classical.py:201

Same signatures, same side effects (``A.rho`` caching, the in-place index sort of ``get_diagonal``, the global
NumPy random stream consumed by the start vector), same return formats.  The sparse products reproduce SciPy's
accumulation order AND its emission order (rows in reverse order of first touch, exact zeros dropped), so a product or
a difference is the array SciPy would have produced, entry for entry -- which matters: the next level's aggregation
walks the stored order (amg_core/smoothed_aggregation.h:185-199).  The spectral radius differs from the reference's
by the rounding of its dot products (different summation order), about 1e-15 relative, and P inherits that.

``device_setup(pyamg)`` patches a reference package the caller hands in (this package never imports it) so that
``pyamg.smoothed_aggregation_solver`` runs those pieces here.  No CPU fallback: what is not supported raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from warnings import warn

import numpy as np
import scipy.sparse as sp
from scipy.linalg import eig

from . import _capi as capi
from .hierarchy import sparse_op

__all__ = ["DeviceCSR", "symmetric_strength_of_connection", "standard_aggregation", "fit_candidates", "approximate_spectral_radius", "jacobi_prolongation_smoother",
           "richardson_prolongation_smoother", "galerkin_product", "device_setup", "device_products"]


def _i32(a):
    a = np.asarray(a)
    if a.dtype != np.int32:
        if a.size and (a.max() > np.iinfo(np.int32).max):
            raise NotImplementedError("index arrays beyond int32 are not supported on the device path")
        a = a.astype(np.int32)
    return np.ascontiguousarray(a)


class DeviceCSR:
    """A scalar fp64 CSR matrix in HBM (``pamg_csr_t``)."""

    def __init__(self, handle, keep=None):
        self.handle = handle
        self._keep = keep                   # the DeviceMatrix a view points into

    @classmethod
    def from_scipy(cls, M) -> "DeviceCSR":
        if not sp.issparse(M):
            raise NotImplementedError(f"operand of type {type(M).__name__} is not a scipy sparse matrix")
        if M.dtype != np.float64:
            raise NotImplementedError(f"setup operators on the device are float64 only (got {M.dtype})")
        if M.format == "bsr" and tuple(M.blocksize) != (1, 1):
            # the flattened scalar view (block after block, row-major inside a block) is SciPy's bsr_matmat order
            from .multilevel import DeviceMatrix
            dm = DeviceMatrix(sparse_op(M))
            h = C.c_void_p()
            capi.check(capi.lib().pamg_csr_view(C.byref(h), dm.handle), "pamg_csr_view")
            return cls(h, keep=dm)
        if M.format not in ("csr", "bsr"):
            M = M.tocsr()
        indptr, indices, data = _i32(M.indptr), _i32(M.indices), np.ascontiguousarray(M.data, dtype=np.float64).reshape(-1)
        h = C.c_void_p()
        capi.check(capi.lib().pamg_csr_create(C.byref(h), M.shape[0], M.shape[1], capi.ptr(indptr), capi.ptr(indices),
                                              capi.ptr(data)), "pamg_csr_create")
        return cls(h)

    @classmethod
    def view_of(cls, dm) -> "DeviceCSR":
        h = C.c_void_p()
        capi.check(capi.lib().pamg_csr_view(C.byref(h), dm.handle), "pamg_csr_view")
        return cls(h, keep=dm)

    @property
    def info(self):
        a = (C.c_int64 * 4)()
        capi.check(capi.lib().pamg_csr_info(self.handle, a), "pamg_csr_info")
        return int(a[0]), int(a[1]), int(a[2])

    @property
    def shape(self):
        m, n, _ = self.info
        return (m, n)

    def matmat(self, other: "DeviceCSR", col_block=1, keep_zeros=False) -> "DeviceCSR":
        """``self @ other`` as SciPy computes and stores it; col_block / keep_zeros: scalar view of a BSR product"""
        h = C.c_void_p()
        capi.check(capi.lib().pamg_csr_matmat(self.handle, other.handle, int(col_block), int(bool(keep_zeros)), C.byref(h)),
                   "pamg_csr_matmat")
        return DeviceCSR(h)

    __matmul__ = matmat

    def __sub__(self, other: "DeviceCSR") -> "DeviceCSR":
        h = C.c_void_p()
        capi.check(capi.lib().pamg_csr_subtract(self.handle, other.handle, C.byref(h)), "pamg_csr_subtract")
        return DeviceCSR(h)

    def to_scipy(self, blocksize=None):
        m, n, nnz = self.info
        indptr = np.empty(m + 1, dtype=np.int32)
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz, dtype=np.float64)
        capi.check(capi.lib().pamg_csr_download(self.handle, capi.ptr(indptr), capi.ptr(indices), capi.ptr(data)),
                   "pamg_csr_download")
        if blocksize is not None and tuple(blocksize) == (1, 1):
            M = sp.bsr_array((data.reshape(-1, 1, 1), indices, indptr), shape=(m, n))
        else:
            M = sp.csr_array((data, indices, indptr), shape=(m, n))
            if blocksize is not None:
                M = M.tobsr(blocksize=tuple(blocksize))
        return M

    def free(self):
        if getattr(self, "handle", None):
            try:
                capi.load().pamg_csr_destroy(self.handle)
            except Exception:       # pragma: no cover
                pass
            self.handle = None
        self._keep = None

    def __del__(self):
        self.free()


# --------------------------------------------------------------------------- spectral radius
ARNOLDI_MAX_BASIS = 31          # pamg_arnoldi_create: basis vectors a device Arnoldi process holds (PAMG_E_ARG beyond)


def _set_tol(dtype):
    """util/params.py set_tol"""
    c = np.dtype(dtype).char.lower()
    if c == "f":
        return 1e3 * np.finfo(np.single).eps
    if c == "d":
        return 1e6 * np.finfo(np.double).eps
    raise ValueError("Attempting to set a tolerance for an unsupported precision.")


class _Arnoldi:
    def __init__(self, dm, maxiter):
        self.n = dm.shape[0]
        self.m = int(min(self.n, maxiter))
        self.dm = dm
        h = C.c_void_p()
        capi.check(capi.lib().pamg_arnoldi_create(C.byref(h), dm.handle, self.m), "pamg_arnoldi_create")
        self.handle = h
        self.planes = 0

    def run(self, v0, breakdown):
        """one process; v0: host (n, 1) array or None (the vector combine() left on the device)"""
        H = np.zeros((self.m + 1, self.m, 2))
        nc, flag = C.c_int(0), C.c_int(0)
        re = im = None
        if v0 is not None:
            v0 = np.ravel(v0)
            re = np.ascontiguousarray(v0.real, dtype=np.float64)
            im = np.ascontiguousarray(v0.imag, dtype=np.float64) if np.iscomplexobj(v0) else None
            self.planes = 2 if im is not None else 1
        capi.check(capi.lib().pamg_arnoldi_run(self.handle, capi.ptr(re), capi.ptr(im), float(breakdown), capi.ptr(H),
                                               C.byref(nc), C.byref(flag)), "pamg_arnoldi_run")
        Hc = H[..., 0] + 1j * H[..., 1] if self.planes == 2 else np.ascontiguousarray(H[..., 0])
        return Hc, int(nc.value), bool(flag.value)

    def combine(self, coef):
        coef = np.ravel(coef)
        re = np.ascontiguousarray(coef.real, dtype=np.float64)
        im = np.ascontiguousarray(coef.imag, dtype=np.float64) if np.iscomplexobj(coef) else None
        capi.check(capi.lib().pamg_arnoldi_combine(self.handle, re.size, capi.ptr(re), capi.ptr(im)), "pamg_arnoldi_combine")
        if im is not None:
            self.planes = 2

    def vector(self):
        re = np.empty(self.n)
        im = np.empty(self.n) if self.planes == 2 else None
        p = C.c_int(0)
        capi.check(capi.lib().pamg_arnoldi_vector(self.handle, capi.ptr(re), capi.ptr(im), C.byref(p)), "pamg_arnoldi_vector")
        v = re + 1j * im if self.planes == 2 else re
        return v.reshape(-1, 1)

    def free(self):
        if getattr(self, "handle", None):
            try:
                capi.load().pamg_arnoldi_destroy(self.handle)
            except Exception:       # pragma: no cover
                pass
            self.handle = None

    def __del__(self):
        self.free()


def _spectral_radius(dm, tol, maxiter, restart, v0, want_vector=False):
    """the restart loop of linalg.py:336-363 on a resident operator"""
    arn = _Arnoldi(dm, maxiter)
    breakdown = _set_tol(np.float64)
    try:
        ev = evect = None
        max_index = 0
        for j in range(restart + 1):
            H, nvecs, breakdown_flag = arn.run(v0 if j == 0 else None, breakdown)
            ev, evect = eig(H[:nvecs, :nvecs], left=False, right=True)
            max_index = np.abs(ev).argmax()
            error = H[nvecs, nvecs - 1] * evect[-1, max_index] if nvecs < H.shape[0] else 0.0
            arn.combine(evect[:, max_index])
            if np.abs(error) / np.abs(ev[max_index]) < tol:
                break
            if breakdown_flag:
                warn(f"Breakdown occurred in step {j}")
                break
        rho = np.abs(ev[max_index])
        return (rho, arn.vector()) if want_vector else rho
    finally:
        arn.free()


def approximate_spectral_radius(A, tol=0.01, maxiter=15, restart=5, symmetric=None, initial_guess=None,
                                return_vector=False):
    """pyamg.util.linalg.approximate_spectral_radius (linalg.py:255-370) with the Arnoldi processes on the device."""
    if not hasattr(A, "rho") or return_vector:
        symmetric = False                                   # linalg.py:311 -- the restart needs the whole basis
        if maxiter < 1:
            raise ValueError("expected maxiter > 0")
        if restart < 0:
            raise ValueError("expected restart >= 0")
        if A.dtype == int:
            raise ValueError("expected A to be float (complex or real)")
        if A.shape[0] != A.shape[1]:
            raise ValueError("expected square A")
        if np.iscomplexobj(A) or (sp.issparse(A) and A.dtype != np.float64):
            raise NotImplementedError(f"approximate_spectral_radius on the device is float64 only (got {A.dtype})")
        if min(A.shape[0], int(maxiter)) > ARNOLDI_MAX_BASIS:
            raise NotImplementedError(f"approximate_spectral_radius on the device holds at most {ARNOLDI_MAX_BASIS} basis vectors "
                                      f"(maxiter={maxiter})")
        if initial_guess is None:
            v0 = np.random.rand(A.shape[1], 1)              # the reference's draw from the global stream
        else:
            if initial_guess.shape[0] != A.shape[0]:
                raise ValueError("initial_guess and A must have same shape")
            if (len(initial_guess.shape) > 1) and (initial_guess.shape[1] > 1):
                raise ValueError("initial_guess must be an (n,1) or (n,) vector")
            v0 = initial_guess.reshape(-1, 1)
            v0 = np.array(v0, dtype=A.dtype)
        from .multilevel import DeviceMatrix
        M = A if sp.issparse(A) else sp.csr_array(np.asarray(A, dtype=np.float64))
        dm = DeviceMatrix(sparse_op(M))
        try:
            out = _spectral_radius(dm, tol, maxiter, restart, v0, want_vector=return_vector)
        finally:
            dm.free()
        rho = out[0] if return_vector else out
        if sp.issparse(A):
            A.rho = rho
        return out if return_vector else rho
    return A.rho


# --------------------------------------------------------------------------- prolongation smoothing
def _diag_inv(S):
    """util/utils.py get_diagonal(S, inv=True): sorts S's indices in place, like the reference does"""
    _sort_indices(S)
    D = S.diagonal()
    Dinv = np.zeros_like(D)
    mask = D != 0.0
    Dinv[mask] = 1.0 / D[mask]
    return Dinv


def _sort_indices(S):
    """S.sort_indices() on the host threads (csrc/pamg_renumber.hip pamg_csr_sort_rows): SciPy sorts the 63 M entries of the 256^3
    hierarchy's first Galerkin product on one core; same arrays afterwards"""
    if S.format not in ("csr", "bsr") or S.has_sorted_indices or S.dtype not in (np.float64, np.float32) \
            or S.indices.dtype != np.int32 or S.indptr.dtype != np.int32 or not (S.indices.flags.c_contiguous and S.data.flags.c_contiguous):
        S.sort_indices()
        return
    block = int(S.blocksize[0] * S.blocksize[1]) if S.format == "bsr" else 1
    nrows = S.indptr.size - 1
    capi.check(capi.load().pamg_csr_sort_rows(capi.dtype_code(S.dtype), nrows, capi.ptr(S.indptr), capi.ptr(S.indices), capi.ptr(S.data), block),
               "pamg_csr_sort_rows")
    S.has_sorted_indices = True


def _scalar_resident(S):
    """S as a resident operator whose values may be rescaled in place (block operators: through their scalar view)"""
    from .multilevel import DeviceMatrix
    if S.format == "bsr" and (S.blocksize[0] != S.blocksize[1] or S.blocksize[0] > 8):
        raise NotImplementedError("prolongation smoothing on the device takes square blocks of at most 8")
    return DeviceMatrix(sparse_op(S))


def _smooth(W, T, degree, product_weight=None, wblock=1):
    """P = T; degree times: P = P - (W @ P)  [richardson: P - product_weight * (W @ P)], W resident.  ``wblock``: the
    (square) block size of W.  With true blocks anywhere SciPy runs bsr_matmat (whole blocks, forward first-touch order)
    and bsr_binop_bsr (a block survives when any entry is non-zero); with 1x1 blocks throughout the CSR routines."""
    tb = tuple(int(v) for v in T.blocksize) if T.format == "bsr" else (1, 1)
    if wblock != tb[0]:
        raise NotImplementedError("prolongation smoothing: the operator's blocks and T's row blocks differ (SciPy re-blocks an operand)")
    blocks = not _all_ones(wblock, wblock, tb[1])
    P = DeviceCSR.from_scipy(T)
    for _ in range(degree):
        U = W.matmat(P, col_block=tb[1], keep_zeros=blocks)
        if product_weight is not None:
            capi.check(capi.lib().pamg_csr_scale(U.handle, float(product_weight)), "pamg_csr_scale")
        if tb == (1, 1):
            Pn = P - U
        else:
            h = C.c_void_p()
            capi.check(capi.lib().pamg_csr_subtract_bsr(P.handle, U.handle, tb[0], tb[1], C.byref(h)), "pamg_csr_subtract_bsr")
            Pn = DeviceCSR(h)
        U.free()
        P.free()
        P = Pn
    out = P.to_scipy(blocksize=T.blocksize if T.format == "bsr" else None)
    P.free()
    return out


def _all_ones(rb, inner, cb):
    return rb == 1 and inner == 1 and cb == 1


def jacobi_prolongation_smoother(S, T, C, B, omega=4.0 / 3.0, degree=1, filter_entries=False, weighting="diagonal"):
    """pyamg.aggregation.smooth.jacobi_prolongation_smoother (smooth.py:61-207): P = (I - omega/rho(D^-1 S) D^-1 S)^degree T,
    with the spectral radius, the scaled operator and the sparse products on the device."""
    if weighting == "block":
        if sp.issparse(S) and S.format == "csr":
            weighting = "diagonal"
        elif sp.issparse(S) and S.format == "bsr":
            if S.blocksize[0] == 1:
                weighting = "diagonal"
        else:
            raise TypeError("S must be sparse BSR or CSR format")
    if filter_entries:
        raise NotImplementedError("jacobi_prolongation_smoother(filter_entries=True) is not on the device path")
    if not (sp.issparse(S) and S.format in ("csr", "bsr")) or not (sp.issparse(T) and T.format in ("csr", "bsr")):
        raise NotImplementedError("jacobi_prolongation_smoother on the device takes CSR / BSR operands")
    if S.dtype != np.float64 or T.dtype != np.float64:
        raise NotImplementedError("jacobi_prolongation_smoother on the device is float64 only")
    if weighting not in ("diagonal", "local", "block"):
        raise ValueError("Incorrect weighting option")
    lib = capi.lib()
    sb = int(S.blocksize[0]) if S.format == "bsr" else 1
    if weighting == "block":
        # smooth.py:165-172: D_inv = the inverted diagonal blocks as a block-diagonal BSR matrix, D_inv_S = D_inv @ S
        # (bsr_matmat with true blocks), scaled by omega / rho(D_inv_S)
        from .relaxation import get_block_diag
        if S.blocksize[0] != S.blocksize[1]:
            raise NotImplementedError("jacobi_prolongation_smoother(weighting='block') takes square blocks")
        D_inv = get_block_diag(S, blocksize=sb, inv_flag=True)
        nb = D_inv.shape[0]
        Dm = sp.bsr_array((D_inv, np.arange(nb, dtype=np.int32), np.arange(nb + 1, dtype=np.int32)), shape=S.shape)
        Dd, Sd = DeviceCSR.from_scipy(Dm), DeviceCSR.from_scipy(S)
        try:
            DS = Dd.matmat(Sd, col_block=sb, keep_zeros=True)
        finally:
            Dd.free()
            Sd.free()
        try:
            M = DS.to_scipy(blocksize=(sb, sb))
        finally:
            DS.free()
        dm = _scalar_resident(M)
        try:
            rho = _spectral_radius(dm, 0.01, 15, 5, np.random.rand(S.shape[1], 1))
            capi.check(lib.pamg_matrix_scale_values(dm.handle, float(omega / rho)), "scale_values")
            W = DeviceCSR.view_of(dm)
            try:
                return _smooth(W, T, degree, wblock=sb)
            finally:
                W.free()
        finally:
            dm.free()
    if weighting == "diagonal":
        D_inv = _diag_inv(S)
        dm = _scalar_resident(S)
        try:
            capi.check(lib.pamg_matrix_scale_rows(dm.handle, capi.ptr(np.ascontiguousarray(D_inv, dtype=np.float64))), "scale_rows")
            rho = _spectral_radius(dm, 0.01, 15, 5, np.random.rand(S.shape[1], 1))
            capi.check(lib.pamg_matrix_scale_values(dm.handle, float(omega / rho)), "scale_values")
            W = DeviceCSR.view_of(dm)
            try:
                return _smooth(W, T, degree, wblock=sb)
            finally:
                W.free()
        finally:
            dm.free()
    if weighting == "local":
        D = np.abs(S) @ np.ones((S.shape[0], 1), dtype=S.dtype)
        D_inv = np.zeros_like(D)
        D_inv[D != 0] = 1.0 / np.abs(D[D != 0])
        dm = _scalar_resident(S)
        try:
            capi.check(lib.pamg_matrix_scale_rows(dm.handle, capi.ptr(np.ascontiguousarray(np.ravel(D_inv), dtype=np.float64))), "scale_rows")
            capi.check(lib.pamg_matrix_scale_values(dm.handle, float(omega)), "scale_values")
            W = DeviceCSR.view_of(dm)
            try:
                return _smooth(W, T, degree, wblock=sb)
            finally:
                W.free()
        finally:
            dm.free()
    raise ValueError("Incorrect weighting option")      # pragma: no cover


def richardson_prolongation_smoother(S, T, omega=4.0 / 3.0, degree=1):
    """pyamg.aggregation.smooth.richardson_prolongation_smoother (smooth.py:210-272): P = P - weight * (S @ P)."""
    if not (sp.issparse(S) and S.format in ("csr", "bsr")) or not (sp.issparse(T) and T.format in ("csr", "bsr")):
        raise NotImplementedError("richardson_prolongation_smoother on the device takes CSR / BSR operands")
    if S.dtype != np.float64 or T.dtype != np.float64:
        raise NotImplementedError("richardson_prolongation_smoother on the device is float64 only")
    weight = omega / approximate_spectral_radius(S)
    Sd = DeviceCSR.from_scipy(S)
    try:
        return _smooth(Sd, T, degree, product_weight=weight,   # weight * (S @ P): the product's values scaled, then P - that
                       wblock=int(S.blocksize[0]) if S.format == "bsr" else 1)
    finally:
        Sd.free()


# --------------------------------------------------------------------------- strength of connection
def symmetric_strength_of_connection(A, theta=0):
    """pyamg.strength.symmetric_strength_of_connection (strength.py:248-348): the strong connections
    ``|a_ij| >= theta sqrt(|a_ii| |a_jj|)`` in A's stored order, as magnitudes scaled by each row's largest.  CSR
    operators go through ``pamg_csr_strength_symmetric``; a BSR operator is reduced to its block pattern / block
    Frobenius norms on the host first, as the reference does."""
    if theta < 0:
        raise ValueError("expected a positive theta")
    if sp.issparse(A) and A.format == "bsr":
        M, N = A.shape
        R, Cb = A.blocksize
        if R != Cb:
            raise ValueError("matrix must have square blocks")
        if theta == 0:
            # every block is a strong connection; |1| scaled by the row maximum 1 stays 1
            return sp.csr_array((np.ones(len(A.indices), dtype=A.dtype), A.indices.copy(), A.indptr.copy()),
                                shape=(M // R, N // Cb))
        norms = np.sqrt((np.conjugate(A.data) * A.data).reshape(-1, R * Cb).sum(axis=1))
        return symmetric_strength_of_connection(sp.csr_array((norms, A.indices, A.indptr), shape=(M // R, N // Cb)), theta)
    if not (sp.issparse(A) and A.format == "csr"):
        raise TypeError("expected CSR or BSR sparse format")
    if A.dtype != np.float64:
        raise NotImplementedError(f"symmetric_strength_of_connection on the device is float64 only (got {A.dtype})")
    if A.shape[0] != A.shape[1]:
        raise NotImplementedError("symmetric_strength_of_connection on the device takes square operators")
    Ad = DeviceCSR.from_scipy(A)
    try:
        h = C.c_void_p()
        capi.check(capi.lib().pamg_csr_strength_symmetric(Ad.handle, float(theta), C.byref(h)), "pamg_csr_strength_symmetric")
        S = DeviceCSR(h)
        out = S.to_scipy()
        _keep_resident(out, S)                               # standard_aggregation(out) comes next in the SA setup
        return out
    finally:
        Ad.free()


# --------------------------------------------------------------------------- aggregation, tentative prolongator
def standard_aggregation(C):
    """pyamg.aggregation.aggregate.standard_aggregation (aggregate.py:12-96): the greedy aggregation of the strength graph
    on the device (``pamg_standard_aggregation``: the reference's three passes, the sequential first one ordered by per-node
    turn counters -- the same aggregates with the same numbers, the same C-points).  Returns (AggOp, Cpts) like the
    reference.  A strength matrix that ``symmetric_strength_of_connection`` of this module just produced is still resident
    and is not shipped again."""
    if not sp.issparse(C) or C.format != "csr":
        raise TypeError("expected csr_array")
    if C.shape[0] != C.shape[1]:
        raise ValueError("expected square matrix")
    index_type = C.indptr.dtype
    n = C.shape[0]
    Tj = np.empty(n, dtype=np.int32)
    Cpts = np.empty(n, dtype=np.int32)
    import ctypes
    na = ctypes.c_int(0)
    lib = capi.lib()
    dev = _resident_copy(C)
    if dev is not None:
        capi.check(lib.pamg_csr_standard_aggregation(dev.handle, capi.ptr(Tj), capi.ptr(Cpts), ctypes.byref(na)),
                   "pamg_csr_standard_aggregation")
        _drop_resident(C)
    else:
        indptr, indices = _i32(C.indptr), _i32(C.indices)
        capi.check(lib.pamg_standard_aggregation(n, capi.ptr(indptr), indptr.size, capi.ptr(indices), indices.size,
                                                 capi.ptr(Tj), n, capi.ptr(Cpts), n, ctypes.byref(na)), "pamg_standard_aggregation")
    num_aggregates = int(na.value)
    Tj = Tj.astype(index_type, copy=False)
    Cpts = Cpts[:num_aggregates].astype(index_type, copy=False)
    if num_aggregates == 0:                                  # aggregate.py:76-79
        return sp.csr_array((n, 1), dtype=np.int32), np.array([], dtype=index_type)
    shape = (n, num_aggregates)
    if Tj.min() == -1:                                       # aggregate.py:84-89: some nodes not aggregated
        mask = Tj != -1
        row = np.arange(n, dtype=index_type)[mask]
        col = Tj[mask]
        data = np.ones(len(col), dtype=np.int32)
        return sp.coo_array((data, (row, col)), shape=shape).tocsr(), Cpts
    Tp = np.arange(n + 1, dtype=index_type)
    Tx = np.ones(len(Tj), dtype=np.int32)
    return sp.csr_array((Tx, Tj, Tp), shape=shape), Cpts


def fit_candidates(AggOp, B, tol=1e-10):
    """pyamg.aggregation.tentative.fit_candidates (tentative.py:9-152): the tentative prolongator Q and the coarse
    candidates R with ``Q @ R = B`` on the aggregates, ``Q.T @ Q = I`` -- per aggregate the reference's modified
    Gram-Schmidt (amg_core.fit_candidates, smoothed_aggregation.h:484-610) on the device, one lane per aggregate, the sums
    in the reference's order.  The blocks come back in AggOp's row order, which is what ``Q.T.tobsr()`` of the reference
    stores, so no transposes are formed on the host."""
    if not sp.issparse(AggOp) or AggOp.format != "csr":
        raise TypeError("expected csr_array for argument AggOp")
    B = np.asarray(B)
    if B.dtype not in ("float32", "float64", "complex64", "complex128"):
        B = np.asarray(B, dtype="float64")
    if len(B.shape) != 2:
        raise ValueError("expected 2d array for argument B")
    if B.shape[0] % AggOp.shape[0] != 0:
        raise ValueError(f"Dimensions of AggOp {AggOp.shape} and B {B.shape} are incompatible")
    if B.dtype.kind == "c":
        raise NotImplementedError("fit_candidates on the device is real only")
    N_fine, N_coarse = AggOp.shape
    K1 = int(B.shape[0] / N_fine)
    K2 = B.shape[1]
    if AggOp.nnz and int(np.diff(AggOp.indptr).max()) > 1:
        raise NotImplementedError("fit_candidates on the device takes an aggregation (one aggregate per node)")
    Tp, Tj = _i32(AggOp.indptr), _i32(AggOp.indices)
    Bc = np.ascontiguousarray(B)
    R = np.empty((N_coarse, K2, K2), dtype=B.dtype)
    Qx = np.empty((AggOp.nnz, K1, K2), dtype=B.dtype)
    fn = capi.lib().pamg_fit_tentative_f64 if B.dtype == np.float64 else capi.lib().pamg_fit_tentative_f32
    capi.check(fn(N_fine, N_coarse, K1, K2, capi.ptr(Tp), capi.ptr(Tj), capi.ptr(Bc), capi.ptr(Qx), capi.ptr(R),
                  float(tol)), "pamg_fit_tentative")
    # the reference builds Q^T as BSR over the CSC arrays and transposes it (tentative.py:146-148): the same matrix
    Q = sp.bsr_array((Qx, Tj.astype(AggOp.indices.dtype, copy=False), Tp.astype(AggOp.indptr.dtype, copy=False)),
                     shape=(K1 * N_fine, K2 * N_coarse))
    return Q, R.reshape(-1, K2)


# a device copy handed from one setup step to the next (strength -> aggregation) without a round trip through the host
_RESIDENT = {}
_HANDOFF = [False]                  # only inside device_setup(): there the next step is known to follow at once


def _clear_resident():
    for ent in _RESIDENT.values():
        ent[0].free()
    _RESIDENT.clear()


def _keep_resident(M, dev):
    _clear_resident()                                        # one hand-off at a time: nothing piles up in HBM
    if not _HANDOFF[0]:
        dev.free()
        return
    _RESIDENT[id(M)] = (dev, M.indices.ctypes.data, M.indptr.ctypes.data, M.nnz)


def _resident_copy(M):
    ent = _RESIDENT.get(id(M))
    if ent is None:
        return None
    dev, pj, pp, nnz = ent
    if M.indices.ctypes.data != pj or M.indptr.ctypes.data != pp or M.nnz != nnz or dev.handle is None:
        return None
    return dev


def _drop_resident(M):
    ent = _RESIDENT.pop(id(M), None)
    if ent is not None:
        ent[0].free()


# --------------------------------------------------------------------------- Galerkin product
def _block(M):
    return tuple(int(v) for v in M.blocksize) if M.format == "bsr" else (1, 1)


def _keeps_zeros(rb, inner, cb):
    """does SciPy's ``bsr_matmat`` store this product block-wise (forward first-touch order, zeros kept)?  Only the
    all-ones shape R == N == C == 1 is forwarded to ``csr_matmat``."""
    return not (rb == 1 and inner == 1 and cb == 1)


def galerkin_product(R, A, P):
    """``R @ A @ P`` (aggregation.py:425, classical.py:201) with both sparse products on the device, in SciPy's
    association -- (R @ A) @ P -- accumulation order and stored order.  The result has the format SciPy's expression
    gives it: that of R (BSR with blocks (R's block rows, P's block columns), or CSR).  Operand combinations for which
    SciPy would first re-block an operand raise."""
    for M in (R, A, P):
        if not sp.issparse(M) or M.dtype != np.float64 or M.format not in ("csr", "bsr"):
            raise NotImplementedError("galerkin_product on the device takes float64 CSR / BSR operands")
    if R.format == "csr":
        if A.format != "csr" or P.format != "csr":
            raise NotImplementedError("galerkin_product: a CSR restriction with BSR operands (SciPy converts and re-sorts them)")
        rb = ab = cb = 1
    else:
        (rb, rc), (ar, ab), (pr, cb) = _block(R), _block(A), _block(P)
        if (A.format == "bsr" and ar != rc) or (A.format == "csr" and rc != 1) or \
           (P.format == "bsr" and pr != ab) or (P.format == "csr" and ab != 1):
            raise NotImplementedError("galerkin_product: block sizes that make SciPy re-block an operand")
    Rd, Ad, Pd = (DeviceCSR.from_scipy(M) for M in (R, A, P))
    # SciPy's bsr_matmat takes the csr_matmat shortcut (reverse first-touch order, exact zeros dropped) only when the
    # row block, the INNER block and the column block are all 1 (sparsetools/bsr.h); every other shape stores whole
    # blocks in forward first-touch order -- R(1,3) @ A(3,3) and (RA)(1,3) @ P(3,1) included
    RA = Rd.matmat(Ad, col_block=ab, keep_zeros=_keeps_zeros(rb, rc if R.format == "bsr" else 1, ab))
    Ac = RA.matmat(Pd, col_block=cb, keep_zeros=_keeps_zeros(rb, ab, cb))
    out = Ac.to_scipy(blocksize=(rb, cb) if R.format == "bsr" else None)
    for d in (RA, Ac, Rd, Ad, Pd):
        d.free()
    return out


# --------------------------------------------------------------------------- SciPy's sparse @ sparse
def _device_product(self, other):
    """``self @ other`` for two float64 CSR operands, or a BSR left operand with a right operand SciPy would not have to
    re-block -- the array SciPy's ``_matmul_sparse`` returns, computed on the device.  NotImplementedError otherwise."""
    if not (sp.issparse(other) and self.ndim == 2 and other.ndim == 2 and self.dtype == np.float64 and other.dtype == np.float64):
        raise NotImplementedError
    if min(self.shape) == 0 or min(other.shape) == 0 or self.nnz == 0 or other.nnz == 0:
        raise NotImplementedError
    if self.format == "csr":
        if other.format != "csr":
            raise NotImplementedError
        rb = cb = n = 1
    elif self.format == "bsr":
        rb, n = (int(v) for v in self.blocksize)
        if other.format == "bsr":
            if int(other.blocksize[0]) != n:
                raise NotImplementedError
            cb = int(other.blocksize[1])
        elif other.format == "csr" and n == 1:
            cb = 1
        else:
            raise NotImplementedError
    else:
        raise NotImplementedError
    Ad, Bd = DeviceCSR.from_scipy(self), DeviceCSR.from_scipy(other)
    try:
        Cd = Ad.matmat(Bd, col_block=cb, keep_zeros=_keeps_zeros(rb, n, cb))
        try:
            M = Cd.to_scipy(blocksize=(rb, cb) if self.format == "bsr" else None)
        finally:
            Cd.free()
    finally:
        Ad.free()
        Bd.free()
    if self.format == "bsr":
        return self._bsr_container((M.data, M.indices, M.indptr), shape=M.shape)
    return self.__class__((M.data, M.indices, M.indptr), shape=M.shape)


def _device_transpose(self, axes=None, copy=False):
    """``self.T`` of a float CSR / BSR array as SciPy builds it (bsr_transpose / csr_tocsc: the same index and value
    arrays, order included), on the device.  NotImplementedError: anything SciPy should keep doing itself."""
    if axes is not None and axes != (1, 0):
        raise NotImplementedError
    if self.dtype not in (np.float64, np.float32) or self.nnz < (1 << 20) or self.indices.dtype != np.int32:
        raise NotImplementedError                       # small operands: the host is as fast as the round trip
    M, N = self.shape
    if self.format == "bsr":
        R, Cb = (int(v) for v in self.blocksize)
    else:
        raise NotImplementedError                       # csr.T is a zero-copy view as CSC in SciPy: nothing to speed up
    nbr, nbc = M // R, N // Cb
    nblk = int(self.indptr[-1])
    indptr = np.empty(nbc + 1, dtype=np.int32)
    indices = np.empty(nblk, dtype=np.int32)
    data = np.empty((nblk, Cb, R), dtype=self.dtype)
    Ap, Aj = _i32(self.indptr), _i32(self.indices)
    Ax = np.ascontiguousarray(self.data).reshape(-1)
    fn = capi.lib().pamg_bsr_transpose_f64 if self.dtype == np.float64 else capi.lib().pamg_bsr_transpose_f32
    capi.check(fn(nbr, nbc, R, Cb, capi.ptr(Ap), capi.ptr(Aj), capi.ptr(Ax), capi.ptr(indptr), capi.ptr(indices), capi.ptr(data)),
               "pamg_bsr_transpose")
    return self._bsr_container((data, indices, indptr), shape=(N, M), copy=copy)


@contextlib.contextmanager
def device_products():
    """While the block runs, ``A @ B`` of two float64 CSR arrays (or BSR @ BSR/CSR without re-blocking) is computed by
    ``pamg_csr_matmat`` -- SciPy's own result, array for array -- through a wrapper around the classes' private
    ``_matmul_sparse`` hook; every other combination takes SciPy's code as before.  This is how the inline
    ``R @ A @ P`` of the reference's setup (aggregation.py:425) reaches the device without editing the reference."""
    saved = []
    for cls in (sp.csr_array, sp.csr_matrix, sp.bsr_array, sp.bsr_matrix):
        orig = cls._matmul_sparse
        own = cls.__dict__.get("_matmul_sparse")

        def wrapper(self, other, _orig=orig):
            try:
                return _device_product(self, other)
            except NotImplementedError:
                return _orig(self, other)
        saved.append((cls, "_matmul_sparse", own))
        cls._matmul_sparse = wrapper
    # P.T of a BSR prolongator (R = P.T, aggregation.py:394-397): bsr_transpose is serial host code in SciPy
    for cls in (sp.bsr_array, sp.bsr_matrix):
        orig = cls.transpose
        own = cls.__dict__.get("transpose")

        def twrapper(self, axes=None, copy=False, _orig=orig):
            try:
                return _device_transpose(self, axes, copy)
            except NotImplementedError:
                return _orig(self, axes=axes, copy=copy)
        saved.append((cls, "transpose", own))
        cls.transpose = twrapper
    try:
        yield
    finally:
        for cls, attr, own in saved:
            if own is None:
                delattr(cls, attr)
            else:
                setattr(cls, attr, own)


# --------------------------------------------------------------------------- patching a reference package
def _device_or_reference(device_fn, reference_fn):
    """a patched setup function: the device twin, and the reference function that was patched out for the inputs the
    device path does not take (it says so with NotImplementedError)"""
    def patched(*args, **kwargs):
        try:
            return device_fn(*args, **kwargs)
        except NotImplementedError:
            return reference_fn(*args, **kwargs)
    patched.__name__ = getattr(reference_fn, "__name__", "patched")
    return patched


def _rho_or_reference(reference_fn):
    """the patched ``approximate_spectral_radius``: float64 sparse matrices go to the device; anything else the reference
    accepts -- a LinearOperator whose matvec is host code (rho_block_D_inv_A), dense arrays, float32 / complex -- stays
    with the reference function that was patched out"""
    def approximate_spectral_radius_(A, *args, **kwargs):
        maxiter = kwargs.get("maxiter", args[1] if len(args) > 1 else 15)
        if sp.issparse(A) and A.dtype == np.float64 and A.shape[0] == A.shape[1] and \
                min(A.shape[0], int(maxiter)) <= ARNOLDI_MAX_BASIS and \
                (A.format != "bsr" or (A.blocksize[0] == A.blocksize[1] and A.blocksize[0] <= 8)):
            return approximate_spectral_radius(A, *args, **kwargs)
        return reference_fn(A, *args, **kwargs)
    return approximate_spectral_radius_


@contextlib.contextmanager
def device_setup(pyamg, prolongation=True, products=True, aggregation=False):
    """Run the setup pieces above inside a reference package the CALLER imported::

        with pyamg_amd.aggregation.device_setup(pyamg):
            ml = pyamg.smoothed_aggregation_solver(A)

    Patched while the block runs: the prolongation smoothers seen by ``pyamg.aggregation.aggregation`` and every
    by-name import of ``approximate_spectral_radius`` (prolongation smoothing, the rho(D^-1 A) of the Jacobi /
    Chebyshev smoother setup).  The Galerkin product is an inline expression in the reference (aggregation.py:425):
    ``products=True`` runs the block under ``device_products()`` so its two sparse products reach the device too
    (INTEGRATION.md shows the one-line change that routes it to ``galerkin_product`` instead).  ``prolongation=False`` leaves the
    prolongation smoothers alone and patches the spectral radius only.  The tentative prolongator (``fit_candidates``) is
    patched too.  ``aggregation=True`` also routes ``standard_aggregation`` to the device: the same aggregates, integer for
    integer -- but the reference's greedy pass costs ~6 ns per node on one host core, while the device version is a
    topological traversal of launch-latency-bound rounds (measured: 0.08 vs 0.05 s at 8 M nodes, 0.24 vs 0.03 s on a
    30-entries-per-row SA level), so it is off by default and meant for pipelines that keep the strength matrix in HBM."""
    import importlib
    targets = []
    for mod, name, fn in (("aggregation.aggregation", "jacobi_prolongation_smoother", jacobi_prolongation_smoother),
                          ("aggregation.aggregation", "richardson_prolongation_smoother", richardson_prolongation_smoother),
                          ("aggregation.aggregation", "symmetric_strength_of_connection", symmetric_strength_of_connection),
                          ("strength", "symmetric_strength_of_connection", symmetric_strength_of_connection),
                          ("aggregation.aggregation", "standard_aggregation", standard_aggregation),
                          ("aggregation.aggregation", "fit_candidates", fit_candidates),
                          ("aggregation.smooth", "approximate_spectral_radius", approximate_spectral_radius),
                          ("relaxation.smoothing", "approximate_spectral_radius", approximate_spectral_radius),
                          ("relaxation.chebyshev", "approximate_spectral_radius", approximate_spectral_radius),
                          ("util.linalg", "approximate_spectral_radius", approximate_spectral_radius)):
        try:
            m = importlib.import_module(f"{pyamg.__name__}.{mod}")
        except ImportError:     # pragma: no cover
            continue
        if not prolongation and name.endswith("prolongation_smoother"):
            continue
        if not aggregation and name == "standard_aggregation":
            continue
        if hasattr(m, name):
            old = getattr(m, name)
            targets.append((m, name, old))
            if name == "approximate_spectral_radius":
                setattr(m, name, _rho_or_reference(old))
            elif name in ("standard_aggregation", "fit_candidates"):
                setattr(m, name, _device_or_reference(fn, old))
            else:
                setattr(m, name, fn)
    was = _HANDOFF[0]
    _HANDOFF[0] = bool(aggregation)             # strength -> aggregation hand-off of the device copy
    try:
        with (device_products() if products else contextlib.nullcontext()):
            yield
    finally:
        _HANDOFF[0] = was
        _clear_resident()
        for m, name, old in targets:
            setattr(m, name, old)
