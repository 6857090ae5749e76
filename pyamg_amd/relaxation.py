"""GPU twins of ``pyamg.relaxation.relaxation`` (same names, arguments, in-place semantics
and error behaviour; reference: pyamg/relaxation/relaxation.py).

``jacobi``, ``gauss_seidel``, ``sor``, ``polynomial``, ``block_jacobi`` and
``block_gauss_seidel`` take SciPy sparse ``A`` and NumPy ``x`` (updated IN PLACE) / ``b``,
exactly like the reference, stage them through HBM and run the HIP kernels of the resident
engine (C ABI Layer 2).  They exist so the reference's own relaxation tests read unchanged
against the device path; inside a solve the hierarchy stays resident instead
(``DeviceMultilevelSolver``).  No CPU fallback.
"""
from __future__ import annotations

from warnings import warn

import numpy as np
from scipy import sparse

from . import _capi as capi
from .hierarchy import sparse_op
from .multilevel import DeviceMatrix

__all__ = ["make_system", "jacobi", "gauss_seidel", "sor", "polynomial", "block_jacobi",
           "block_gauss_seidel", "jacobi_indexed", "gauss_seidel_indexed", "schwarz", "schwarz_parameters", "cf_jacobi", "fc_jacobi", "cf_block_jacobi", "fc_block_jacobi", "gauss_seidel_ne",
           "gauss_seidel_nr", "jacobi_ne"]


def _as_format(A, formats):
    """A in one of the wanted storage formats (first one preferred); ``['csr']`` alone keeps the reference's
    habit of warning when something that is neither CSR nor BSR has to be converted."""
    have = A.format if sparse.issparse(A) else None
    if formats is None or have in formats:
        return A
    if list(formats) == ["csr"]:
        if have == "bsr":
            return A.tocsr()
        warn("implicit conversion to CSR", sparse.SparseEfficiencyWarning)
        return sparse.csr_array(A)
    return sparse.csr_array(A).asformat(formats[0])


def _system_defects(A, x, b):
    """(exception type, message) for everything the relaxation entry points refuse, in the order the
    reference reports them (relaxation.py:59-97)."""
    for name, v in (("x", x), ("b", b)):
        if not isinstance(v, np.ndarray):
            yield ValueError, f"expected numpy array for argument {name}"
            return
    rows, cols = A.shape
    if rows != cols:
        yield ValueError, "expected square matrix"
    ok_shapes = ((rows,), (rows, 1))
    for name, v in (("x", x), ("b", b)):
        if v.shape not in ok_shapes:
            yield ValueError, f"{name} has invalid dimensions"
    if len({np.dtype(A.dtype), x.dtype, b.dtype}) != 1:
        yield TypeError, "arguments A, x, and b must have the same dtype"
    if not x.flags.c_contiguous or not x.flags.aligned or not x.flags.writeable:
        yield ValueError, "x must be contiguous in memory"


def make_system(A, x, b, formats=None):
    """A (in an accepted format), flat views of x and b -- or the exception the reference's ``make_system``
    raises for the same arguments (relaxation.py:15-97): ValueError for non-square A, wrong shapes, non-array
    or non-contiguous x; TypeError for mixed dtypes.  x is returned as a VIEW: the sweeps update it in place."""
    A = _as_format(A, formats)
    for exc, msg in _system_defects(A, x, b):
        raise exc(msg)
    return A, x.reshape(-1), b.reshape(-1)


class _Staged:
    """A, x, b in HBM for the duration of one call; x is copied back on exit."""

    def __init__(self, A, x, b, work=0):
        self.x_host = x
        self.A = DeviceMatrix(sparse_op(A))
        self.x = capi.DeviceArray.from_host(x)
        self.b = capi.DeviceArray.from_host(b)
        self.work = capi.DeviceArray(max(1, work * x.size), x.dtype) if work else None

    def finish(self):
        capi.sync()
        self.x_host[:] = self.x.download()
        for d in (self.x, self.b, self.work):
            if d is not None:
                d.free()
        self.A.free()


def _square_blocks(A):
    if sparse.issparse(A) and A.format == "bsr":
        R, C = A.blocksize
        if R != C:
            raise ValueError("BSR blocks must be square")


def sor(A, x, b, omega, iterations=1, sweep="forward"):
    """SOR on Ax=b, in place (reference: relaxation.py:100-154)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _square_blocks(A)
    if sweep not in ("forward", "backward", "symmetric"):
        raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')
    st = _Staged(A, x, b)
    st.A.gauss_seidel(st.x, st.b, sweep=sweep, omega=float(omega), iterations=iterations)
    st.finish()


def gauss_seidel(A, x, b, iterations=1, sweep="forward", omega=1.0):
    """Gauss-Seidel on Ax=b, in place (reference: relaxation.py:265-346; the reference
    wrapper has the same private ``omega`` keyword, used by ``sor``)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _square_blocks(A)
    if sweep not in ("forward", "backward", "symmetric"):
        raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')
    st = _Staged(A, x, b)
    st.A.gauss_seidel(st.x, st.b, sweep=sweep, omega=float(omega), iterations=iterations)
    st.finish()


def jacobi(A, x, b, iterations=1, omega=1.0):
    """Weighted Jacobi on Ax=b, in place (reference: relaxation.py:349-420)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _square_blocks(A)
    if A.shape[0] <= 0:
        return
    st = _Staged(A, x, b, work=1)
    st.A.jacobi(st.x, st.b, st.work, float(np.real(omega)), iterations)
    st.finish()


def _indexed_sweeps(A, x, b, plan, iterations, omega):
    """Shared body of jacobi_indexed / cf_jacobi / fc_jacobi: ``plan`` = [(row list, sweeps), ...] run
    ``iterations`` times, each sweep one amg_core.jacobi_indexed (relaxation.h:382-427), device-resident."""
    if sparse.issparse(A) and A.format == "bsr":
        raise NotImplementedError("indexed Jacobi on BSR operators is not on the device path (CSR only)")
    st = _Staged(A, x, b)
    subs = []
    for rows, _ in plan:
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        sub = st.A.subset_rows(rows)
        subs.append((sub, capi.DeviceArray(max(rows.size, 1), st.A.dtype)))
    for _ in range(iterations):
        for (sub, work), (_, sweeps) in zip(subs, plan):
            for _ in range(sweeps):
                sub.jacobi_indexed(st.x, st.b, float(np.real(omega)), work)
    st.finish()
    for sub, work in subs:
        sub.free()
        work.free()


def jacobi_indexed(A, x, b, indices, iterations=1, omega=1.0):
    """Weighted Jacobi on the listed rows only, in place (reference: relaxation.py:1077-1138)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    indices = np.asarray(indices, dtype=np.int32)
    if (indices.min(initial=0) < 0) or (indices.max(initial=-1) > A.shape[0] - 1):
        raise ValueError("indices must range from 0, ..., N-1")       # relaxation.py:1108-1109
    _indexed_sweeps(A, x, b, [(indices, 1)], iterations, omega)


def _subdomain_blocks(A, subdomain, subdomain_ptr):
    """amg_core.extract_subblocks (relaxation.h:1332-1418): the dense diagonal block A[S_d, S_d] of every subdomain,
    row-major, one after another (A's rows and every S_d sorted)."""
    sizes = np.diff(subdomain_ptr)
    ptr = np.zeros(subdomain_ptr.shape, dtype=A.indices.dtype)
    ptr[1:] = np.cumsum(sizes * sizes)
    out = np.zeros((ptr[-1],), dtype=A.dtype)
    for d in range(len(sizes)):
        rows = subdomain[subdomain_ptr[d]:subdomain_ptr[d + 1]]
        m = rows.size
        blk = out[ptr[d]:ptr[d + 1]].reshape(m, m)
        for r, row in enumerate(rows):
            cols = A.indices[A.indptr[row]:A.indptr[row + 1]]
            at = np.searchsorted(rows, cols)
            hit = (at < m) & (rows[np.minimum(at, m - 1)] == cols)
            blk[r, at[hit]] = A.data[A.indptr[row]:A.indptr[row + 1]][hit]
    return out, ptr


def schwarz_parameters(A, subdomain=None, subdomain_ptr=None, inv_subblock=None, inv_subblock_ptr=None):
    """The subdomains and inverted diagonal blocks of Schwarz relaxation (reference: relaxation.py:1002-1075): by default
    one subdomain per row, its sparsity pattern; every block inverted by LAPACK's gelss with the reference's rank
    tolerance; cached on the matrix as ``A.schwarz_parameters`` like the reference caches it."""
    import scipy.linalg as la
    cached = getattr(A, "schwarz_parameters", None)
    if cached is not None:
        given = subdomain is not None and subdomain_ptr is not None
        if not given or (np.array_equal(cached[0], subdomain) and np.array_equal(cached[1], subdomain_ptr)):
            return cached                                   # same subdomains (or none asked for): what was built before
    if subdomain is None or subdomain_ptr is None:          # default: row i's subdomain = the columns of row i
        subdomain, subdomain_ptr = A.indices.copy(), A.indptr.copy()
    if inv_subblock is None or inv_subblock_ptr is None:
        inv_subblock, inv_subblock_ptr = _subdomain_blocks(A, subdomain, subdomain_ptr)
        single = np.dtype(A.dtype).char.lower() == "f"
        rank_tol = (1e3 * np.finfo(np.single).eps) if single else (1e6 * np.finfo(np.double).eps)    # util/params.py set_tol
        gelss, = la.get_lapack_funcs(["gelss"], (np.ones((1,), dtype=A.dtype),))
        for d, m in enumerate(np.diff(subdomain_ptr)):
            blk = inv_subblock[inv_subblock_ptr[d]:inv_subblock_ptr[d + 1]]
            # pseudo-inverse of the block: least-squares solve against the identity, like the reference does it
            blk[:] = np.ravel(gelss(blk.reshape(m, m), np.eye(m, m, dtype=A.dtype), cond=rank_tol, overwrite_a=True, overwrite_b=True)[1])
    A.schwarz_parameters = (subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr)
    return A.schwarz_parameters


def schwarz(A, x, b, iterations=1, subdomain=None, subdomain_ptr=None, inv_subblock=None, inv_subblock_ptr=None,
            sweep="forward"):
    """Multiplicative overlapping Schwarz, in place (reference: relaxation.py:157-262)."""
    from . import amg_core
    A, x, b = make_system(A, x, b, formats=["csr"])
    A.sort_indices()
    if subdomain is None and inv_subblock is not None:
        raise ValueError("inv_subblock must be None if subdomain is None")
    subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr = schwarz_parameters(A, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr)
    nsub = subdomain_ptr.shape[0] - 1
    if sweep == "forward":
        row_start, row_stop, row_step = 0, nsub, 1
    elif sweep == "backward":
        row_start, row_stop, row_step = nsub - 1, -1, -1
    elif sweep == "symmetric":
        for _ in range(iterations):
            schwarz(A, x, b, 1, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr, "forward")
            schwarz(A, x, b, 1, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr, "backward")
        return
    else:
        raise ValueError("valid sweep directions: 'forward', 'backward', and 'symmetric'")
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)       # noqa: E731
    Ap, Aj, Sp, Sj, Tp = i32(A.indptr), i32(A.indices), i32(subdomain_ptr), i32(subdomain), i32(inv_subblock_ptr)
    Tx = np.ascontiguousarray(inv_subblock, dtype=A.dtype)
    for _ in range(iterations):
        amg_core.overlapping_schwarz_csr(Ap, Aj, A.data, x, b, Tx, Tp, Sj, Sp, nsub, A.shape[0], row_start, row_stop, row_step)


def gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="forward"):
    """Gauss-Seidel on the listed rows, in the listed order, in place (reference: relaxation.py:662-731; a row may be
    listed more than once; 'backward' walks the list from its end, 'symmetric' = forward then backward)."""
    from . import amg_core
    A, x, b = make_system(A, x, b, formats=["csr"])
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    if indices.size and (indices.min() < 0 or indices.max() > A.shape[0] - 1):
        raise ValueError("indices must range from 0, ..., N-1")
    if sweep == "forward":
        row_start, row_stop, row_step = 0, len(indices), 1
    elif sweep == "backward":
        row_start, row_stop, row_step = len(indices) - 1, -1, -1
    elif sweep == "symmetric":
        for _ in range(iterations):
            gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="forward")
            gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="backward")
        return
    else:
        raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')
    Ap, Aj = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)
    for _ in range(iterations):
        amg_core.gauss_seidel_indexed(Ap, Aj, A.data, x, b, indices, row_start, row_stop, row_step)


def cf_jacobi(A, x, b, Cpts, Fpts, iterations=1, f_iterations=1, c_iterations=1, omega=1.0):
    """CF Jacobi, in place: per iteration c_iterations sweeps over Cpts, then f_iterations over Fpts
    (reference: relaxation.py:1141-1203)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _indexed_sweeps(A, x, b, [(Cpts, c_iterations), (Fpts, f_iterations)], iterations, omega)


def fc_jacobi(A, x, b, Cpts, Fpts, iterations=1, f_iterations=1, c_iterations=1, omega=1.0):
    """FC Jacobi, in place: per iteration f_iterations sweeps over Fpts, then c_iterations over Cpts
    (reference: relaxation.py:1206-1268)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _indexed_sweeps(A, x, b, [(Fpts, f_iterations), (Cpts, c_iterations)], iterations, omega)


def _ne_spec(kind, A, iterations, sweep, omega):
    from .hierarchy import _normal_equation_spec
    return _normal_equation_spec(kind, A, iterations, sweep, omega)


def gauss_seidel_ne(A, x, b, iterations=1, sweep="forward", omega=1.0, Dinv=None):
    """Kaczmarz relaxation (Gauss-Seidel on A A^H y = b, x = A^H y), in place
    (reference: relaxation.py:815-901)."""
    A, x, b = make_system(A, x, b, formats=["csr"])
    if sweep not in ("forward", "backward", "symmetric"):
        raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')
    sm = _ne_spec("gauss_seidel_ne", A, iterations, sweep, omega)
    dD = capi.DeviceArray.from_host(np.ascontiguousarray(sm.Dinv if Dinv is None else np.ravel(Dinv), dtype=A.dtype))
    st = _Staged(A, x, b)
    st.A.kaczmarz(st.x, dD, float(omega), sweep, iterations, b=st.b)
    st.finish()
    dD.free()


def gauss_seidel_nr(A, x, b, iterations=1, sweep="forward", omega=1.0, Dinv=None):
    """Gauss-Seidel on A^H A x = A^H b, in place (reference: relaxation.py:904-988): the residual is formed
    once per directional call, then swept ``iterations`` times; 'symmetric' = forward call + backward call."""
    from .multilevel import DeviceMatrix
    A, x, b = make_system(A, x, b, formats=["csc", "csr"])
    if sweep not in ("forward", "backward", "symmetric"):
        raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')
    sm = _ne_spec("gauss_seidel_nr", A, iterations, sweep, omega)
    dD = capi.DeviceArray.from_host(np.ascontiguousarray(sm.Dinv if Dinv is None else np.ravel(Dinv), dtype=A.dtype))
    st = _Staged(A, x, b, work=1)
    At = DeviceMatrix(sm.At)
    Ar = DeviceMatrix(sm.Ar) if sm.Ar is not None else st.A        # rows sorted = the CSC product's summation order

    def call(direction, its):
        Ar.spmv(capi.SPMV_RESID, st.x, st.work, b=st.b)                      # r = b - A x  (:983)
        At.kaczmarz(st.work, dD, float(omega), direction, its, xout=st.x)

    if sweep == "symmetric":
        for _ in range(iterations):
            call("forward", 1)
            call("backward", 1)
    else:
        call(sweep, iterations)
    st.finish()
    At.free()
    if Ar is not st.A:
        Ar.free()
    dD.free()


def jacobi_ne(A, x, b, iterations=1, omega=1.0):
    """Jacobi on the normal equations A A^H y = b, x = A^H y, in place (reference: relaxation.py:741-812)."""
    from .multilevel import DeviceMatrix
    A, x, b = make_system(A, x, b, formats=["csr"])
    sm = _ne_spec("jacobi_ne", A, iterations, "forward", omega)
    dD = capi.DeviceArray.from_host(np.ascontiguousarray(sm.Dinv, dtype=A.dtype))
    st = _Staged(A, x, b, work=1)
    At = DeviceMatrix(sm.At)                                               # (omega A)^T, row-oriented
    for _ in range(iterations):
        st.A.spmv(capi.SPMV_RESID, st.x, st.work, b=st.b)                    # r = b - A x
        capi.check(capi.lib().pamg_vec_mul(capi.dtype_code(A.dtype), x.size, st.work.ptr, dD.ptr, st.work.ptr, None),
                   "pamg_vec_mul")                                       # delta = r .* Dinv
        At.spmv(capi.SPMV_ACC, st.work, st.x)                              # x += (omega A)^T delta
    st.finish()
    At.free()
    dD.free()


def polynomial(A, x, b, coefficients, iterations=1):
    """x += p(A) (b - A x) with Horner evaluation, in place (reference: relaxation.py:585-659)."""
    A, x, b = make_system(A, x, b, formats=None)
    if not sparse.issparse(A):
        A = sparse.csr_array(A)
    st = _Staged(A, x, b, work=3)
    x_is_zero = bool(np.linalg.norm(x) == 0)          # the reference's own test (:649)
    st.A.polynomial(st.x, st.b, st.work, np.asarray(coefficients, dtype=np.float64), iterations, x_is_zero)
    st.finish()


def get_block_diag(A, blocksize, inv_flag=True):
    """The block diagonal of A as an (N/blocksize, blocksize, blocksize) array, inverted block by block when
    ``inv_flag`` (reference: util/utils.py:603-692 -- same result, same caching on ``A.block_D_inv`` / ``A.block_D``).
    The pseudo-inverses are ``amg_core.pinv_array`` on the device (``pamg_pinv_array``: the reference's one-sided Jacobi
    SVD per block, bit for bit); blocks of 7 and more go through LAPACK in the reference and are not on the device path."""
    if not sparse.issparse(A):
        raise TypeError("Expected sparse matrix")
    if A.shape[0] != A.shape[1]:
        raise ValueError("Expected square matrix")
    if np.mod(A.shape[0], blocksize) != 0:
        raise ValueError("blocksize and A.shape must be compatible")
    nb = A.shape[0] // blocksize
    cached = getattr(A, "block_D_inv" if inv_flag else "block_D", None)
    if cached is not None and cached.shape == (nb, blocksize, blocksize):
        return cached
    if A.format != "bsr" or tuple(A.blocksize) != (blocksize, blocksize):
        A = A.tobsr(blocksize=(blocksize, blocksize))
    if A.dtype.kind == "c":
        raise NotImplementedError("get_block_diag of a complex operator is not on the device path")      # (asfptype keeps complex; a cast would drop the imaginary part)
    if A.dtype.kind != "f":
        A = A.astype(np.float64)
    # the stored block at (i, i); with duplicates SciPy's diagonal() of the position matrix adds the positions up -- the
    # reference inherits that; canonical operators have one
    rows = np.repeat(np.arange(nb), np.diff(A.indptr))
    on = np.flatnonzero(A.indices == rows)
    block_diag = np.zeros((nb, blocksize, blocksize), dtype=A.dtype)
    if np.unique(rows[on]).size != on.size:
        raise NotImplementedError("get_block_diag: duplicate diagonal blocks are not on the device path")
    block_diag[rows[on]] = A.data[on]
    if not inv_flag:
        A.block_D = block_diag
        return block_diag
    if blocksize >= 7:
        raise NotImplementedError("get_block_diag(inv_flag=True) with blocks of 7 and more (LAPACK in the reference) is not on the device path")
    from . import amg_core
    amg_core.pinv_array(block_diag.ravel(), nb, blocksize, "T")
    A.block_D_inv = block_diag
    return block_diag


def _block_prep(A, blocksize, Dinv):
    A = A.tobsr(blocksize=(blocksize, blocksize))       # relaxation.py:475,556
    if Dinv is None:
        Dinv = get_block_diag(A, blocksize=blocksize, inv_flag=True)     # relaxation.py:477-478
    if Dinv.shape[0] != int(A.shape[0] / blocksize):
        raise ValueError("Dinv and A have incompatible dimensions")
    if (Dinv.shape[1] != blocksize) or (Dinv.shape[2] != blocksize):
        raise ValueError("Dinv and blocksize are incompatible")
    return A, np.ascontiguousarray(Dinv, dtype=A.dtype)


def _check_unit_dinv(A, Dinv):
    """argument checks of the block wrappers for blocksize 1 (relaxation.py:479-484).  A caller-supplied Dinv is honoured
    by the reference (it multiplies by Dinv[i] whatever it holds); the point kernels used here for 1x1 blocks divide by
    a_ii, so a Dinv that is NOT the inverse of the diagonal is refused instead of silently ignored."""
    if Dinv is not None:
        Dinv = np.asarray(Dinv)
        if Dinv.shape[0] != A.shape[0]:
            raise ValueError("Dinv and A have incompatible dimensions")
        if Dinv.ndim != 3 or Dinv.shape[1] != 1 or Dinv.shape[2] != 1:
            raise ValueError("Dinv and blocksize are incompatible")
        d = np.asarray(A.diagonal(), dtype=np.float64)
        want = np.where(d != 0, 1.0 / np.where(d != 0, d, 1.0), 0.0)
        if not np.allclose(np.ravel(Dinv), want, rtol=1e-12, atol=0):
            raise NotImplementedError("blocksize=1 with a Dinv that is not 1 / diag(A) is not on the device path")


def block_jacobi(A, x, b, Dinv=None, blocksize=1, iterations=1, omega=1.0):
    """Block Jacobi, in place (reference: relaxation.py:423-499)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    if blocksize == 1:
        # 1x1 blocks: the point method, like the reference's own smoother setup substitutes it (smoothing.py:565-569).  A
        # direct call of the reference multiplies by the pseudo-inverse of a_ii where jacobi divides by a_ii: the two
        # can differ in the last bit per sweep -- the one place where this module is not bit-identical
        _check_unit_dinv(A, Dinv)
        return jacobi(A.tocsr(), x, b, iterations=iterations, omega=omega)
    A, Dinv = _block_prep(A, blocksize, Dinv)
    st = _Staged(A, x, b, work=1)
    dD = capi.DeviceArray.from_host(Dinv.reshape(-1))
    st.A.block_jacobi(st.x, st.b, st.work, dD, float(np.real(omega)), iterations)
    st.finish()
    dD.free()


def _indexed_block_sweeps(A, x, b, plan, Dinv, blocksize, iterations, omega):
    """Shared body of cf_block_jacobi / fc_block_jacobi: ``plan`` = [(block-row list, sweeps), ...] run ``iterations``
    times, each sweep one amg_core.block_jacobi_indexed (relaxation.h:1129-1199), device-resident."""
    if blocksize == 1:
        # 1x1 blocks: the point methods (smoothing.py:731-734, 768-771); see block_jacobi for the last-bit caveat
        _check_unit_dinv(A, Dinv)
        return _indexed_sweeps(A.tocsr(), x, b, plan, iterations, omega)
    A, Dinv = _block_prep(A, blocksize, Dinv)
    nb = A.shape[0] // blocksize
    lists = []
    for rows, _ in plan:
        rows = np.asarray(rows, dtype=np.int64)
        if rows.size and (rows.min() < 0 or rows.max() > nb - 1):
            raise ValueError("block indices must range from 0, ..., N/blocksize - 1")
        lists.append(np.ascontiguousarray((rows[:, None] * blocksize + np.arange(blocksize)).ravel(), dtype=np.int32))
    st = _Staged(A, x, b, work=1)
    dD = capi.DeviceArray.from_host(Dinv.reshape(-1))
    dI = [capi.DeviceArray.from_host(idx) if idx.size else None for idx in lists]
    lib = capi.lib()
    for _ in range(iterations):
        for idx, d, (_, sweeps) in zip(lists, dI, plan):
            for _ in range(sweeps):
                if d is not None:
                    capi.check(lib.pamg_matrix_block_jacobi_indexed(st.A.handle, dD.ptr, st.x.ptr, st.b.ptr, d.ptr, idx.size,
                                                                    float(np.real(omega)), st.work.ptr, None),
                               "pamg_matrix_block_jacobi_indexed")
    st.finish()
    dD.free()
    for d in dI:
        if d is not None:
            d.free()


def cf_block_jacobi(A, x, b, Cpts, Fpts, Dinv=None, blocksize=1, iterations=1, f_iterations=1, c_iterations=1, omega=1.0):
    """CF block Jacobi, in place: per iteration c_iterations sweeps over the C block rows, then f_iterations over the F
    block rows (reference: relaxation.py:1271-1340)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _indexed_block_sweeps(A, x, b, [(Cpts, c_iterations), (Fpts, f_iterations)], Dinv, blocksize, iterations, omega)


def fc_block_jacobi(A, x, b, Cpts, Fpts, Dinv=None, blocksize=1, iterations=1, f_iterations=1, c_iterations=1, omega=1.0):
    """FC block Jacobi, in place: F block rows first, then C (reference: relaxation.py:1342-1411)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    _indexed_block_sweeps(A, x, b, [(Fpts, f_iterations), (Cpts, c_iterations)], Dinv, blocksize, iterations, omega)


def block_gauss_seidel(A, x, b, iterations=1, sweep="forward", blocksize=1, Dinv=None):
    """Block Gauss-Seidel, in place (reference: relaxation.py:502-582)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    if sweep not in ("forward", "backward", "symmetric"):
        raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')
    if blocksize == 1:
        # 1x1 blocks: the point method (smoothing.py:595-599); see block_jacobi for the last-bit caveat
        _check_unit_dinv(A, Dinv)
        return gauss_seidel(A.tocsr(), x, b, iterations=iterations, sweep=sweep)
    A, Dinv = _block_prep(A, blocksize, Dinv)
    st = _Staged(A, x, b)
    dD = capi.DeviceArray.from_host(Dinv.reshape(-1))
    st.A.block_gauss_seidel(st.x, st.b, dD, sweep, iterations)
    st.finish()
    dD.free()
