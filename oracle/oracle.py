"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference solve phase: ctypes bindings of ``liboracle.so``
(``amg_oracle.c``) plus a NumPy restatement of the reference's Python-side logic
(smoother wrappers, cycle recursion, outer iteration).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``pyamg_amd/``) never does.

Parity status: PINNED (tests/test_oracle.py): against the reference's known answers
(tests/golden/known_answers.json), against the reference's only numeric fixture
(docs/paper/example.res.txt, 22 residual norms) and bit-for-bit against the real
reference (oracle/_ref) on seeded inputs.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None

_I = C.c_int
_P = C.c_void_p


def build(force: bool = False) -> Path:
    so = HERE / "liboracle.so"
    srcs = [HERE / "amg_oracle.c", HERE / "amg_oracle_impl.h"]
    if force or not so.exists() or so.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(HERE), "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
    return _LIB


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _sfx(dt):
    dt = np.dtype(dt)
    if dt == np.float64:
        return "f64", C.c_double
    if dt == np.float32:
        return "f32", C.c_float
    raise TypeError(f"oracle: unsupported dtype {dt}")


def _call(name, dt, *args):
    sfx, _ = _sfx(dt)
    fn = getattr(lib(), f"orc_{name}_{sfx}")
    fn.restype = None
    fn(*args)


# ----------------------------------------------------------------- kernels (array level)
def csr_matvec(n_row, Ap, Aj, Ax, x, y):
    """y += A x (SciPy csr_matvec semantics)."""
    _call("csr_matvec", Ax.dtype, _I(n_row), _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y))


def bsr_matvec(n_brow, R, Cc, Ap, Aj, Ax, x, y):
    _call("bsr_matvec", Ax.dtype, _I(n_brow), _I(R), _I(Cc), _ptr(Ap), _ptr(Aj), _ptr(Ax),
          _ptr(x), _ptr(y))


def gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step):
    _call("gauss_seidel", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b),
          _I(row_start), _I(row_stop), _I(row_step))


def sor_gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, omega):
    _, ct = _sfx(Ax.dtype)
    _call("sor_gauss_seidel", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b),
          _I(row_start), _I(row_stop), _I(row_step), ct(omega))


def bsr_gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, blocksize):
    _call("bsr_gauss_seidel", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b),
          _I(row_start), _I(row_stop), _I(row_step), _I(blocksize))


def jacobi(Ap, Aj, Ax, x, b, temp, row_start, row_stop, row_step, omega):
    _, ct = _sfx(Ax.dtype)
    _call("jacobi", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b), _ptr(temp),
          _I(row_start), _I(row_stop), _I(row_step), ct(omega))


def jacobi_indexed(Ap, Aj, Ax, x, b, indices, omega):
    """amg_core.jacobi_indexed (relaxation.h:382-427)."""
    _, ct = _sfx(Ax.dtype)
    temp = np.empty_like(x)
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    _call("jacobi_indexed", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _I(x.size), _ptr(b), _ptr(idx),
          _I(idx.size), _ptr(temp), ct(omega))


def overlapping_schwarz_csr(Ap, Aj, Ax, x, b, Tx, Tp, Sj, Sp, row_start, row_stop, row_step):
    """amg_core.overlapping_schwarz_csr (relaxation.h:1420-1492)."""
    Tp, Sj, Sp = (np.ascontiguousarray(a, dtype=np.int32) for a in (Tp, Sj, Sp))
    Tx = np.ascontiguousarray(Tx, dtype=Ax.dtype)
    maxsize = int(np.diff(Sp).max(initial=1))
    work = np.zeros(2 * maxsize, dtype=Ax.dtype)
    _call("overlapping_schwarz_csr", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b), _ptr(Tx), _ptr(Tp), _ptr(Sj),
          _ptr(Sp), _I(row_start), _I(row_stop), _I(row_step), _ptr(work), _I(maxsize))


def relax_schwarz(A, x, b, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr, iterations=1, sweep="forward"):
    """relaxation.py:157-262 (A: CSR SparseOp with sorted rows = lvl.Acsr)."""
    nsub = len(subdomain_ptr) - 1
    if sweep == "symmetric":
        for _ in range(iterations):
            relax_schwarz(A, x, b, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr, 1, "forward")
            relax_schwarz(A, x, b, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr, 1, "backward")
        return
    r = (0, nsub, 1) if sweep == "forward" else (nsub - 1, -1, -1)
    for _ in range(iterations):
        overlapping_schwarz_csr(A.indptr, A.indices, A.data, x, b, inv_subblock, inv_subblock_ptr, subdomain, subdomain_ptr, *r)


def gauss_seidel_indexed(Ap, Aj, Ax, x, b, Id, row_start, row_stop, row_step):
    """amg_core.gauss_seidel_indexed (relaxation.h:736-790)."""
    idx = np.ascontiguousarray(Id, dtype=np.int32)
    _call("gauss_seidel_indexed", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b), _ptr(idx),
          _I(row_start), _I(row_stop), _I(row_step))


def relax_gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="forward"):
    """relaxation.py:662-731 (CSR)."""
    m = len(indices)
    if sweep == "symmetric":
        for _ in range(iterations):
            relax_gauss_seidel_indexed(A, x, b, indices, 1, "forward")
            relax_gauss_seidel_indexed(A, x, b, indices, 1, "backward")
        return
    r = (0, m, 1) if sweep == "forward" else (m - 1, -1, -1)
    for _ in range(iterations):
        gauss_seidel_indexed(A.indptr, A.indices, A.data, x, b, indices, *r)


def block_jacobi_indexed(Ap, Aj, Ax, x, b, Dinv, indices, omega, blocksize):
    """amg_core.block_jacobi_indexed (relaxation.h:1129-1199)."""
    _, ct = _sfx(Ax.dtype)
    temp = np.empty_like(x)
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    _call("block_jacobi_indexed", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _I(x.size), _ptr(b), _ptr(Dinv),
          _ptr(idx), _I(idx.size), _ptr(temp), ct(omega), _I(blocksize))


def gauss_seidel_ne(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, Dinv, omega):
    """amg_core.gauss_seidel_ne (relaxation.h:875-904)."""
    _, ct = _sfx(Ax.dtype)
    _call("gauss_seidel_ne", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b), _I(row_start), _I(row_stop),
          _I(row_step), _ptr(Dinv), ct(omega))


def gauss_seidel_nr(Ap, Aj, Ax, x, z, col_start, col_stop, col_step, Dinv, omega):
    """amg_core.gauss_seidel_nr (relaxation.h:939-975); Ap/Aj/Ax = CSC arrays of A."""
    _, ct = _sfx(Ax.dtype)
    _call("gauss_seidel_nr", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(z), _I(col_start), _I(col_stop),
          _I(col_step), _ptr(Dinv), ct(omega))


def jacobi_ne(Ap, Aj, Ax, x, delta, temp, row_start, row_stop, row_step, omega):
    """amg_core.jacobi_ne (relaxation.h:811-840)."""
    _, ct = _sfx(Ax.dtype)
    _call("jacobi_ne", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(delta), _ptr(temp), _I(row_start),
          _I(row_stop), _I(row_step), ct(omega))


def bsr_jacobi(Ap, Aj, Ax, x, b, temp, row_start, row_stop, row_step, blocksize, omega):
    _, ct = _sfx(Ax.dtype)
    _call("bsr_jacobi", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b), _ptr(temp),
          _I(row_start), _I(row_stop), _I(row_step), _I(blocksize), ct(omega))


def block_jacobi(Ap, Aj, Ax, x, b, Dinv, temp, row_start, row_stop, row_step, omega, blocksize):
    _, ct = _sfx(Ax.dtype)
    _call("block_jacobi", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b), _ptr(Dinv),
          _ptr(temp), _I(row_start), _I(row_stop), _I(row_step), ct(omega), _I(blocksize))


def block_gauss_seidel(Ap, Aj, Ax, x, b, Dinv, row_start, row_stop, row_step, blocksize):
    _call("block_gauss_seidel", Ax.dtype, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(b),
          _ptr(Dinv), _I(row_start), _I(row_stop), _I(row_step), _I(blocksize))


# ----------------------------------------------------------------- operator level
def pinv_array(AA, m, n, TransA="T"):
    """in place: every n x n block of AA (m, n, n) replaced by its pseudo-inverse (amg_core.pinv_array, linalg.h:930-1000)"""
    assert AA.flags.c_contiguous and AA.size == m * n * n
    _call("pinv_array", AA.dtype, _ptr(AA), _I(m), _I(n), C.c_char(TransA.encode()))


def standard_aggregation(Ap, Aj):
    """(x, y, count): amg_core.standard_aggregation (smoothed_aggregation.h:137-268)"""
    n = Ap.size - 1
    x = np.empty(n, dtype=np.int32)
    y = np.empty(max(n, 1), dtype=np.int32)
    fn = lib().orc_standard_aggregation
    fn.restype = C.c_int
    cnt = fn(_I(n), _ptr(np.ascontiguousarray(Ap, dtype=np.int32)), _ptr(np.ascontiguousarray(Aj, dtype=np.int32)), _ptr(x), _ptr(y))
    return x, y[:cnt], int(cnt)


def fit_candidates(n_col, K1, K2, Ap, Ai, B, tol):
    """(Ax, R): amg_core.fit_candidates (smoothed_aggregation.h:484-610) on the CSC arrays of AggOp"""
    B = np.ascontiguousarray(B)
    Ax = np.empty((Ai.size, K1, K2), dtype=B.dtype)
    R = np.empty((n_col, K2, K2), dtype=B.dtype)
    sfx, ct = _sfx(B.dtype)
    fn = getattr(lib(), f"orc_fit_candidates_{sfx}")
    fn.restype = None
    fn(_I(n_col), _I(K1), _I(K2), _ptr(np.ascontiguousarray(Ap, dtype=np.int32)), _ptr(np.ascontiguousarray(Ai, dtype=np.int32)),
       _ptr(Ax), _ptr(B), _ptr(R), ct(tol))
    return Ax, R


def matvec(op, x):
    """``op @ x`` for a SparseOp-like (fmt, shape, blocksize, indptr, indices, data);
    fresh zero-initialised result as SciPy's ``_matmul_vector`` does."""
    y = np.zeros(op.shape[0], dtype=np.result_type(op.data.dtype, x.dtype))
    x = np.ascontiguousarray(x, dtype=y.dtype)
    R, Cc = op.blocksize
    if op.fmt == "csr":
        csr_matvec(op.shape[0], op.indptr, op.indices, op.data, x, y)
    else:
        bsr_matvec(op.shape[0] // R, R, Cc, op.indptr, op.indices, op.data, x, y)
    return y


def _sweep_bounds(sweep, nrows):
    if sweep == "forward":
        return 0, nrows, 1
    if sweep == "backward":
        return nrows - 1, -1, -1
    raise ValueError('valid sweep directions: "forward", "backward", and "symmetric"')


def relax_gauss_seidel(A, x, b, iterations=1, sweep="forward", omega=1.0):
    """relaxation.py:265-346 (incl. its quirks: 'symmetric' drops omega, BSR ignores omega)."""
    R = A.blocksize[0]
    if A.fmt == "bsr" and A.blocksize[0] != A.blocksize[1]:
        raise ValueError("BSR blocks must be square")
    if sweep == "symmetric":
        for _ in range(iterations):
            relax_gauss_seidel(A, x, b, 1, "forward")
            relax_gauss_seidel(A, x, b, 1, "backward")
        return
    bs = 1 if A.fmt == "csr" else R
    r0, r1, rs = _sweep_bounds(sweep, len(x) // bs)
    for _ in range(iterations):
        if A.fmt == "csr":
            if omega != 1.0:
                sor_gauss_seidel(A.indptr, A.indices, A.data, x, b, r0, r1, rs, omega)
            else:
                gauss_seidel(A.indptr, A.indices, A.data, x, b, r0, r1, rs)
        else:
            bsr_gauss_seidel(A.indptr, A.indices, A.data, x, b, r0, r1, rs, R)


def relax_sor(A, x, b, omega, iterations=1, sweep="forward"):
    """relaxation.py:100-154."""
    for _ in range(iterations):
        relax_gauss_seidel(A, x, b, 1, sweep, omega)


def relax_jacobi(A, x, b, iterations=1, omega=1.0):
    """relaxation.py:349-420 (always a full forward sweep; temp allocated per call)."""
    n = A.shape[0]
    if n <= 0:
        return
    temp = np.empty_like(x)
    om = A.data.dtype.type(omega)
    for _ in range(iterations):
        if A.fmt == "csr":
            jacobi(A.indptr, A.indices, A.data, x, b, temp, 0, n, 1, om)
        else:
            R = A.blocksize[0]
            if A.blocksize[0] != A.blocksize[1]:
                raise ValueError("BSR blocks must be square")
            bsr_jacobi(A.indptr, A.indices, A.data, x, b, temp, 0, n // R, 1, R, om)


def _as_bsr(A, blocksize):
    """relaxation.py:475,556: ``A.tobsr(blocksize)`` on every call."""
    if A.fmt == "bsr" and A.blocksize == (blocksize, blocksize):
        return A
    from types import SimpleNamespace
    M = A.to_scipy().tobsr(blocksize=(blocksize, blocksize))
    return SimpleNamespace(fmt="bsr", shape=M.shape, blocksize=(blocksize, blocksize),
                           indptr=M.indptr.astype(np.int32), indices=M.indices.astype(np.int32),
                           data=np.ascontiguousarray(M.data).reshape(-1))


def relax_block_jacobi(A, x, b, Dinv, blocksize, iterations=1, omega=1.0):
    """relaxation.py:423-499."""
    A = _as_bsr(A, blocksize)
    nb = A.shape[0] // blocksize
    temp = np.empty_like(x)
    Dflat = np.ascontiguousarray(Dinv).reshape(-1)
    for _ in range(iterations):
        block_jacobi(A.indptr, A.indices, A.data, x, b, Dflat, temp, 0, nb, 1,
                     A.data.dtype.type(omega), blocksize)


def relax_block_gauss_seidel(A, x, b, Dinv, blocksize, iterations=1, sweep="forward"):
    """relaxation.py:502-582."""
    A = _as_bsr(A, blocksize)
    Dflat = np.ascontiguousarray(Dinv).reshape(-1)
    if sweep == "symmetric":
        for _ in range(iterations):
            relax_block_gauss_seidel(A, x, b, Dinv, blocksize, 1, "forward")
            relax_block_gauss_seidel(A, x, b, Dinv, blocksize, 1, "backward")
        return
    r0, r1, rs = _sweep_bounds(sweep, len(x) // blocksize)
    for _ in range(iterations):
        block_gauss_seidel(A.indptr, A.indices, A.data, x, b, Dflat, r0, r1, rs, blocksize)


def relax_polynomial(A, x, b, coefficients, iterations=1):
    """relaxation.py:585-659: Horner evaluation of p(A) r added to x."""
    for _ in range(iterations):
        if np.linalg.norm(x) == 0:
            residual = b
        else:
            residual = b - matvec(A, x)
        h = coefficients[0] * residual
        for c in coefficients[1:]:
            h = c * residual + matvec(A, h)
        x += h


def relax_jacobi_indexed(A, x, b, indices, iterations=1, omega=1.0):
    """relaxation.py:1077-1138 (CSR)."""
    if A.fmt != "csr":
        raise NotImplementedError("oracle: indexed Jacobi is restated for CSR operators")
    om = A.data.dtype.type(omega)
    for _ in range(iterations):
        jacobi_indexed(A.indptr, A.indices, A.data, x, b, indices, om)


def relax_cf_jacobi(A, x, b, Cpts, Fpts, iterations=1, f_iterations=1, c_iterations=1, omega=1.0, f_first=False):
    """relaxation.py:1141-1203 (cf_jacobi) and :1206-1268 (fc_jacobi, f_first=True)."""
    if A.fmt != "csr":
        raise NotImplementedError("oracle: CF/FC Jacobi is restated for CSR operators")
    om = A.data.dtype.type(omega)
    plan = [(Fpts, f_iterations), (Cpts, c_iterations)] if f_first else [(Cpts, c_iterations), (Fpts, f_iterations)]
    for _ in range(iterations):
        for pts, sweeps in plan:
            for _ in range(sweeps):
                jacobi_indexed(A.indptr, A.indices, A.data, x, b, pts, om)


def relax_cf_block_jacobi(A, x, b, Cpts, Fpts, Dinv, blocksize, iterations=1, f_iterations=1, c_iterations=1, omega=1.0,
                          f_first=False):
    """relaxation.py:1271-1340 (cf_block_jacobi) and :1342-1411 (fc_block_jacobi, f_first=True)."""
    A = _as_bsr(A, blocksize)
    om = A.data.dtype.type(omega)
    Dflat = np.ascontiguousarray(Dinv).reshape(-1)
    plan = [(Fpts, f_iterations), (Cpts, c_iterations)] if f_first else [(Cpts, c_iterations), (Fpts, f_iterations)]
    for _ in range(iterations):
        for pts, sweeps in plan:
            for _ in range(sweeps):
                block_jacobi_indexed(A.indptr, A.indices, A.data, x, b, Dflat, pts, om, blocksize)


def relax_gauss_seidel_ne(A, x, b, Dinv, iterations=1, sweep="forward", omega=1.0):
    """relaxation.py:815-901 (Kaczmarz; A: CSR SparseOp, Dinv = 1 / squared row norms)."""
    if sweep == "symmetric":
        for _ in range(iterations):
            relax_gauss_seidel_ne(A, x, b, Dinv, 1, "forward", omega)
            relax_gauss_seidel_ne(A, x, b, Dinv, 1, "backward", omega)
        return
    r0, r1, rs = _sweep_bounds(sweep, len(x))
    for _ in range(iterations):
        gauss_seidel_ne(A.indptr, A.indices, A.data, x, b, r0, r1, rs, Dinv, omega)


def relax_gauss_seidel_nr(A, At, x, b, Dinv, iterations=1, sweep="forward", omega=1.0, Ar=None):
    """relaxation.py:904-988.  A: CSR SparseOp (for r = b - A x), At: CSR SparseOp of A^T = the CSC arrays
    of A, Dinv = 1 / squared column norms.  The residual is formed once per directional call (:983)."""
    if sweep == "symmetric":
        for _ in range(iterations):
            relax_gauss_seidel_nr(A, At, x, b, Dinv, 1, "forward", omega, Ar)
            relax_gauss_seidel_nr(A, At, x, b, Dinv, 1, "backward", omega, Ar)
        return
    c0, c1, cs = _sweep_bounds(sweep, len(x))
    r = b - matvec(A if Ar is None else Ar, x)           # Ar: A with sorted rows (= the CSC product's order)
    for _ in range(iterations):
        gauss_seidel_nr(At.indptr, At.indices, At.data, x, r, c0, c1, cs, Dinv, omega)


def relax_jacobi_ne(A, x, b, Dinv, iterations=1, omega=1.0):
    """relaxation.py:741-812 (A: CSR SparseOp, Dinv = 1 / squared row norms)."""
    n = len(x)
    temp = np.zeros_like(x)
    om = A.data.dtype.type(omega)
    for _ in range(iterations):
        delta = (np.ravel(b - matvec(A, x)) * np.ravel(Dinv)).astype(A.data.dtype)
        jacobi_ne(A.indptr, A.indices, A.data, x, delta, temp, 0, n, 1, om)


# ----------------------------------------------------------------- Krylov methods as smoothers / coarse solvers
def _vnorm(v):
    """util/linalg.py:13-52 norm(): sqrt of the inner product"""
    v = np.ravel(v)
    return np.sqrt(np.inner(v.conj(), v)).real


def krylov_cg(A, b, x, tol, maxiter):
    """krylov/_cg.py:87-200 with M = identity (z IS r), criteria 'rr'; x is updated in place and returned"""
    if not maxiter:
        maxiter = int(1.3 * len(b)) + 2
    r = b - matvec(A, x)
    p = r.copy()
    rz = np.inner(r, r)
    normb = _vnorm(b)
    if normb == 0.0:
        normb = 1.0
    if np.linalg.norm(r) < tol * normb:
        return x
    it = 0
    while True:
        Ap = matvec(A, p)
        rz_old = rz
        pAp = np.inner(Ap, p)
        if pAp < 0.0:
            return x
        alpha = rz / pAp
        x += alpha * p
        if it % 8 and it > 0:
            r -= alpha * Ap
        else:
            r = b - matvec(A, x)
        rz = np.inner(r, r)
        if rz < 0.0:
            return x
        p *= rz / rz_old
        p += r
        it += 1
        if np.linalg.norm(r) < tol * normb or it == maxiter:
            return x


def krylov_cgn(A, At, b, x, tol, maxiter, nr):
    """krylov/_cgne.py:96-210 (nr = False) and krylov/_cgnr.py:96-212 (nr = True), M = identity, criteria 'rr'"""
    n = len(b)
    if not maxiter or maxiter > 1.3 * n:
        maxiter = int(np.ceil(1.3 * n)) + 2
    r = b - matvec(A, x)
    if nr:
        rhat = matvec(At, r)
        p = rhat.copy()
        old_zr = np.inner(rhat, rhat)
    else:
        p = matvec(At, r)
        old_zr = np.inner(r, r)
    normb = _vnorm(b)
    if normb == 0.0:
        normb = 1.0
    if _vnorm(r) < tol * normb:
        return x
    it = 0
    while True:
        if nr:
            w = matvec(A, p)
            alpha = old_zr / np.inner(w, w)
        else:
            alpha = old_zr / np.inner(p, p)
        x += alpha * p
        if it % 8 and it > 0:
            r -= alpha * (w if nr else matvec(A, p))
        else:
            r = b - matvec(A, x)
        if nr:
            rhat = matvec(At, r)
            new_zr = np.inner(rhat, rhat)
        else:
            new_zr = np.inner(r, r)
        beta = new_zr / old_zr
        old_zr = new_zr
        p *= beta
        p += rhat if nr else matvec(At, r)
        it += 1
        if np.linalg.norm(r) < tol * normb or it == maxiter:
            return x


def krylov_gmres(A, b, x, tol, maxiter, restart):
    """krylov/_gmres_householder.py:120-330 with M = identity: Householder reflectors, Givens rotations on the leading
    entries, Horner-scheme update.  NumPy dot products where the reference runs sequential C loops (krylov.h:37-130), so
    this restatement is pinned to the reference to 1e-12, not bit for bit."""
    n = len(b)
    if n < 2:
        raise NotImplementedError("n == 1 is special-cased by the reference")
    if restart:
        max_outer, max_inner = (maxiter if maxiter else 1), min(restart, n)
    else:
        max_outer, max_inner = 1, (min(maxiter, n) if maxiter else min(n, 40))
    m = max_inner
    sign = lambda t: 1.0 if t == 0.0 else t / abs(t)     # noqa: E731
    r = b - matvec(A, x)
    normr = _vnorm(r)
    normb = _vnorm(b)
    if normb == 0.0:
        normb = 1.0
    if normr < tol * normb:
        return x
    for _ in range(max_outer):
        H = np.zeros((m, m))
        g = np.zeros(m + 1)
        Q = []
        W = np.zeros((m, n))
        w = r.copy()
        beta = sign(w[0]) * normr
        w[0] += beta
        w /= _vnorm(w)
        W[0] = w
        g[0] = -beta
        inner = 0
        for inner in range(m):
            v = -2.0 * W[inner][inner] * W[inner]
            v[inner] += 1.0
            for j in range(inner - 1, -1, -1):
                v -= 2.0 * np.dot(W[j], v) * W[j]
            v = matvec(A, v)
            for j in range(inner + 1):
                v -= 2.0 * np.dot(W[j], v) * W[j]
            if inner != n - 1:
                alpha = _vnorm(v[inner + 1:])
                if alpha != 0.0:
                    alpha = sign(v[inner + 1]) * alpha
                    if inner < m - 1:
                        wn = np.zeros(n)
                        wn[inner + 1:] = v[inner + 1:]
                        wn[inner + 1] += alpha
                        wn /= _vnorm(wn)
                        W[inner + 1] = wn
                    v[inner + 1] = -alpha
                    v[inner + 2:] = 0.0
            lead = min(inner + 2, n)
            hv = np.zeros(m + 2)
            hv[:lead] = v[:lead]
            for k, (c, sn) in enumerate(Q):
                t = hv[k]
                hv[k] = c * t + sn * hv[k + 1]
                hv[k + 1] = -sn * t + c * hv[k + 1]
            if inner != n - 1 and hv[inner + 1] != 0.0:
                f, gg = hv[inner], hv[inner + 1]
                if f == 0.0:
                    c, sn = 0.0, 1.0
                else:
                    rr = np.copysign(np.hypot(f, gg), f)
                    c, sn = f / rr, gg / rr
                Q.append((c, sn))
                g[inner], g[inner + 1] = c * g[inner] + sn * g[inner + 1], -sn * g[inner] + c * g[inner + 1]
                hv[inner], hv[inner + 1] = c * f + sn * gg, 0.0
            H[:, inner] = hv[:m]
            if inner < m - 1:
                normr = abs(g[inner + 1])
                if normr < tol * normb:
                    break
        k = min(inner + 1, m)
        y = np.zeros(k)
        for i in range(k - 1, -1, -1):
            y[i] = (g[i] - np.dot(H[i, i + 1:k], y[i + 1:k])) / H[i, i]
        u = np.zeros(n)
        for j in range(k - 1, -1, -1):
            u[j] += y[j]
            u -= 2.0 * np.dot(W[j], u) * W[j]
        x += u
        r = b - matvec(A, x)
        normr = _vnorm(r)
        mx = np.abs(x)
        ok = mx != 0.0
        if ok.any() and np.max(np.abs(u[ok]) / mx[ok]) < 1e-12:
            return x
        if normr < tol * normb:
            return x
    return x


def apply_smoother(s, A, x, b):
    """Dispatch a SmootherSpec exactly as the reference's bound callable would run."""
    if s is None or s.kind == "none":
        return
    if s.kind == "jacobi":
        relax_jacobi(A, x, b, s.iterations, s.omega)
    elif s.kind == "gauss_seidel":
        relax_gauss_seidel(A, x, b, s.iterations, s.sweep)
    elif s.kind == "sor":
        relax_sor(A, x, b, s.omega, s.iterations, s.sweep)
    elif s.kind == "polynomial":
        relax_polynomial(A, x, b, s.coefficients, s.iterations)
    elif s.kind == "block_jacobi":
        relax_block_jacobi(A, x, b, s.Dinv, s.blocksize, s.iterations, s.omega)
    elif s.kind == "block_gauss_seidel":
        relax_block_gauss_seidel(A, x, b, s.Dinv, s.blocksize, s.iterations, s.sweep)
    elif s.kind == "gauss_seidel_ne":
        relax_gauss_seidel_ne(A if getattr(s, "Ar", None) is None else s.Ar, x, b, s.Dinv, s.iterations, s.sweep, s.omega)
    elif s.kind == "gauss_seidel_nr":
        relax_gauss_seidel_nr(A, s.At, x, b, s.Dinv, s.iterations, s.sweep, s.omega, getattr(s, "Ar", None))
    elif s.kind == "jacobi_ne":
        relax_jacobi_ne(A if getattr(s, "Ar", None) is None else s.Ar, x, b, s.Dinv, s.iterations, s.omega)
    elif s.kind in ("cf_jacobi", "fc_jacobi"):
        relax_cf_jacobi(A, x, b, s.Cpts, s.Fpts, s.iterations, s.f_iterations, s.c_iterations, s.omega,
                        f_first=(s.kind == "fc_jacobi"))
    elif s.kind == "schwarz":
        relax_schwarz(A if getattr(s, "Ar", None) is None else s.Ar, x, b, s.subdomain, s.subdomain_ptr, s.inv_subblock,
                      s.inv_subblock_ptr, s.iterations, s.sweep)
    elif s.kind in ("cf_block_jacobi", "fc_block_jacobi"):
        relax_cf_block_jacobi(A, x, b, s.Cpts, s.Fpts, s.Dinv, s.blocksize, s.iterations, s.f_iterations, s.c_iterations,
                              s.omega, f_first=(s.kind == "fc_block_jacobi"))
    elif s.kind == "cg":
        x[:] = krylov_cg(A, b, x.copy(), s.tol, s.iterations)
    elif s.kind in ("cgne", "cgnr"):
        x[:] = krylov_cgn(A, s.At, b, x.copy(), s.tol, s.iterations, s.kind == "cgnr")
    elif s.kind == "gmres":
        x[:] = krylov_gmres(A, b, x.copy(), s.tol, s.iterations, s.restart)
    else:
        raise ValueError(f"oracle: unknown smoother kind {s.kind}")


# ----------------------------------------------------------------- cycle + outer loop
class OracleSolver:
    """NumPy restatement of ``MultilevelSolver.solve`` / ``__solve``
    (multilevel.py:398-582, 584-662) over a HierarchySpec."""

    def __init__(self, spec):
        self.spec = spec

    def coarse_solve(self, b):
        """multilevel.py:717-721 (dense operator applied to b) / :801-803 (nnz == 0)."""
        if self.spec.coarse_kind == "relax":                 # multilevel.py:765-782: sweeps from x = 0
            x = np.zeros_like(b)
            apply_smoother(self.spec.coarse_smoother, self.spec.levels[-1].A, x, b)
            return x
        if self.spec.coarse_kind == "zero":
            return np.zeros(b.shape)
        return np.dot(self.spec.coarse_op, b)

    def cycle(self, lvl, x, b, cycle="V", cycles_per_level=1):
        """multilevel.py:584-662 (V, W, F)."""
        L = self.spec.levels[lvl]
        apply_smoother(L.pre, L.A, x, b)
        residual = b - matvec(L.A, x)
        coarse_b = matvec(L.R, residual)
        coarse_x = np.zeros_like(coarse_b)
        if lvl == len(self.spec.levels) - 2:
            coarse_x[:] = self.coarse_solve(coarse_b)
        elif cycle == "V":
            self.cycle(lvl + 1, coarse_x, coarse_b, "V")
        elif cycle == "W":
            self.cycle(lvl + 1, coarse_x, coarse_b, cycle)
            self.cycle(lvl + 1, coarse_x, coarse_b, cycle)
        elif cycle == "F":
            self.cycle(lvl + 1, coarse_x, coarse_b, cycle, cycles_per_level)
            for _ in range(cycles_per_level):
                self.cycle(lvl + 1, coarse_x, coarse_b, "V", 1)
        elif cycle == "AMLI":
            # multilevel.py:628-656: two search directions per level, each the result of a
            # recursive AMLI solve started from ones, made A_c-orthogonal to the earlier one,
            # then an exact line search; coarse_b is deflated in place between the two.
            Ac = self.spec.levels[lvl + 1].A
            dirs = np.zeros((2, coarse_b.shape[0]), dtype=coarse_b.dtype)
            for k in range(2):
                dirs[k, :] = 1
                self.cycle(lvl + 1, dirs[k, :], coarse_b, cycle)
                for j in range(k):
                    bkj = np.inner(dirs[j], matvec(Ac, dirs[k])) / np.inner(dirs[j], matvec(Ac, dirs[j]))
                    dirs[k, :] -= bkj * dirs[j, :]
                Adk = matvec(Ac, dirs[k, :])
                step = np.inner(dirs[k], coarse_b) / np.inner(dirs[k], Adk)
                coarse_x += step * dirs[k]
                coarse_b -= step * Adk
        else:
            raise TypeError(f"Unrecognized cycle type ({cycle})")
        x += matvec(L.P, coarse_x)
        apply_smoother(L.post, L.A, x, b)

    def solve(self, b, x0=None, tol=1e-5, maxiter=100, cycle="V", residuals=None,
              callback=None, cycles_per_level=1, return_info=False):
        """multilevel.py:398-582, accel=None branch."""
        A = self.spec.levels[0].A
        x = np.zeros_like(b) if x0 is None else np.array(x0)
        cycle = str(cycle).upper()
        normb = np.linalg.norm(b)
        if normb == 0.0:
            normb = 1.0
        tp = np.result_type(b.dtype, x.dtype, A.data.dtype)
        b = np.ravel(np.asarray(b, dtype=tp))
        x = np.ravel(np.asarray(x, dtype=tp))
        normr = np.linalg.norm(b - matvec(A, x))
        if residuals is not None:
            residuals[:] = [normr]
        it = 0
        while True:
            if len(self.spec.levels) == 1:
                x = self.coarse_solve(b)
            else:
                self.cycle(0, x, b, cycle, cycles_per_level)
            it += 1
            normr = np.linalg.norm(b - matvec(A, x))
            if residuals is not None:
                residuals.append(normr)
            if callback is not None:
                callback(x)
            if normr < tol * normb:
                return (x, 0) if return_info else x
            if it == maxiter:
                return (x, it) if return_info else x
