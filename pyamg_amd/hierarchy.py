"""Host-side description of a multigrid hierarchy, ready to ship to HBM once.

The setup phase stays in the unmodified reference on the host (north star); this
module only *reads* a constructed ``pyamg.MultilevelSolver`` -- duck-typed, pyamg
is never imported here -- and normalises it into plain arrays:

* every operator becomes a row-oriented ``SparseOp`` (CSR or BSR, int32 indices,
  flattened values).  The reference's level operators arrive as ``csr_array``
  (Ruge-Stuben; SA level 0), ``bsr_array`` (SA levels >= 1, P, R -- blocksize (1,1)
  for scalar PDEs, ``aggregation/tentative.py:142-152``) or ``csc_array`` (``R`` of a
  hand-built hierarchy, ``multilevel.py:180-182``); see SURVEY.md §8(a3), App. A.5.
* every smoother callable attached by ``change_smoothers`` (``relaxation/smoothing.py:75``)
  becomes a ``SmootherSpec`` whose scalars are READ BACK from the callable (partial
  keywords / closure cells) -- they contain Arnoldi estimates seeded from
  ``np.random`` (``util/linalg.py:336``) and must never be recomputed.
* the coarsest-level solver (``multilevel.py:665-826``) becomes a dense operator
  ``x_c = M b_c`` for the linear direct solvers ('pinv', 'lu', 'cholesky', 'splu').

Anything that cannot be represented raises ``NotImplementedError``: there is no CPU
fallback in the product path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from functools import partial
from typing import List, Optional, Tuple

import numpy as np
import scipy.sparse as sp

__all__ = ["SparseOp", "SmootherSpec", "LevelSpec", "HierarchySpec", "extract",
           "sparse_op", "smoother_spec", "save_spec", "load_spec"]

SUPPORTED_DTYPES = (np.float64, np.float32)


@dataclass
class SparseOp:
    """Row-oriented sparse operator (CSR when blocksize == (1, 1) and fmt == 'csr')."""
    fmt: str                        # 'csr' | 'bsr'  -- the reference's format (selects arithmetic flavour)
    shape: Tuple[int, int]
    blocksize: Tuple[int, int]
    indptr: np.ndarray              # int32 [n_brow + 1]
    indices: np.ndarray             # int32 [n_blocks]
    data: np.ndarray                # T [n_blocks * R * C], blocks row-major
    src_format: str = "csr"         # format the reference level actually held

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def nnz(self) -> int:
        return int(self.data.size)

    @property
    def n_brow(self) -> int:
        return int(self.indptr.size - 1)

    def to_scipy(self):
        R, C = self.blocksize
        if self.fmt == "csr":
            return sp.csr_array((self.data, self.indices, self.indptr), shape=self.shape)
        return sp.bsr_array((self.data.reshape(-1, R, C), self.indices, self.indptr),
                            shape=self.shape)


@dataclass
class SmootherSpec:
    """One pre/post smoother of one level, as the reference would apply it.

    kind: 'jacobi' | 'gauss_seidel' | 'sor' | 'polynomial' | 'block_jacobi' |
          'block_gauss_seidel' | 'cf_jacobi' | 'fc_jacobi' | 'gauss_seidel_ne' |
          'gauss_seidel_nr' | 'jacobi_ne' | 'cf_block_jacobi' | 'fc_block_jacobi' | 'schwarz' | 'none' |
          'cg' | 'gmres' | 'cgne' | 'cgnr' (Krylov methods as smoothers / coarse solvers: iterations = maxiter (0 = the
          method's default), tol, restart (0 = None), At = A^H for cgne / cgnr)
    """
    kind: str
    iterations: int = 1
    omega: float = 1.0
    sweep: str = "forward"
    coefficients: Optional[np.ndarray] = None
    Dinv: Optional[np.ndarray] = None           # (n_brow, bs, bs)
    blocksize: int = 1
    name: str = ""                              # reference-side display name
    Fpts: Optional[np.ndarray] = None           # cf_jacobi / fc_jacobi: int32 row lists
    Cpts: Optional[np.ndarray] = None
    f_iterations: int = 1
    c_iterations: int = 1
    At: Optional["SparseOp"] = None             # gauss_seidel_nr: CSR of A^T (= CSC arrays of A); jacobi_ne: same, values * omega
    Ar: Optional["SparseOp"] = None             # normal-equation smoothers / schwarz: A with sorted rows as the reference's wrapper sees it (None: the level's A is)
    subdomain: Optional[np.ndarray] = None      # schwarz: rows of every subdomain (sorted, unique), int32
    subdomain_ptr: Optional[np.ndarray] = None
    inv_subblock: Optional[np.ndarray] = None   # schwarz: inverted diagonal blocks, row-major, one after another
    inv_subblock_ptr: Optional[np.ndarray] = None
    tol: float = 0.0                            # Krylov smoothers
    restart: int = 0


@dataclass
class LevelSpec:
    A: SparseOp
    P: Optional[SparseOp] = None
    R: Optional[SparseOp] = None
    pre: Optional[SmootherSpec] = None
    post: Optional[SmootherSpec] = None


@dataclass
class HierarchySpec:
    levels: List[LevelSpec] = field(default_factory=list)
    coarse_kind: str = "dense"                  # 'dense' (x = M b) | 'zero' (A_c.nnz == 0) | 'relax' (sweeps from x = 0)
    coarse_op: Optional[np.ndarray] = None      # dense (n_c, n_c), row-major
    coarse_name: str = "'pinv'"
    coarse_smoother: Optional[SmootherSpec] = None   # 'relax': the relaxation method the reference's coarse solver applies
    coarse_host: Optional[tuple] = None         # 'host': (the caller's coarse solver object, its coarsest operator) -- solved on the host inside the cycle

    @property
    def dtype(self):
        return self.levels[0].A.dtype


# --------------------------------------------------------------------------- operators
def _as_int32(a, what):
    a = np.asarray(a)
    if a.size and (a.max(initial=0) > np.iinfo(np.int32).max):
        raise NotImplementedError(f"{what}: index does not fit int32 (reference is int32-only, "
                                  "amg_core/instantiate.yml:2-6)")
    return np.ascontiguousarray(a, dtype=np.int32)


def sparse_op(M) -> SparseOp:
    """Normalise a scipy sparse matrix into a row-oriented SparseOp (no value changes).

    csr -> as is.  bsr -> as is (blocks row-major, flattened).  csc (the lazily
    transposed ``R = P.T`` of hand-built hierarchies) -> CSR with sorted columns, which
    makes the per-row summation order equal to the column-scatter order of SciPy's
    ``csc_matvec``.  Anything else -> CSR.
    """
    if not sp.issparse(M):
        raise NotImplementedError(f"operator of type {type(M).__name__} is not a scipy sparse matrix")
    if M.dtype.type not in SUPPORTED_DTYPES:
        raise NotImplementedError(f"dtype {M.dtype} not supported on device (float64/float32 only)")
    src = M.format
    if src == "bsr":
        R, C = M.blocksize
        return SparseOp("bsr", tuple(M.shape), (int(R), int(C)), _as_int32(M.indptr, "indptr"),
                        _as_int32(M.indices, "indices"),
                        np.ascontiguousarray(M.data).reshape(-1), src)
    if src != "csr":
        M = M.tocsr()
        if src == "csc":
            M.sort_indices()
    return SparseOp("csr", tuple(M.shape), (1, 1), _as_int32(M.indptr, "indptr"),
                    _as_int32(M.indices, "indices"), np.ascontiguousarray(M.data), src)


# --------------------------------------------------------------------------- smoothers
def _closure_vars(fn) -> dict:
    names = getattr(getattr(fn, "__code__", None), "co_freevars", ()) or ()
    cells = getattr(fn, "__closure__", None) or ()
    return {n: c.cell_contents for n, c in zip(names, cells)}


def smoother_spec(fn, A) -> SmootherSpec:
    """Translate one smoother callable ``fn(A, x, b)`` into a SmootherSpec.

    Dispatch is on the *wrapped* function (``partial.func``) or on the closure contents,
    never on ``__name__`` (``update_wrapper`` makes a point smoother advertise itself as
    ``block_gauss_seidel`` when blocksize == 1, ``smoothing.py:595-599``) -- SURVEY §8(b).
    """
    if fn is None:
        return SmootherSpec("none", iterations=0, name="None")
    shown = getattr(fn, "__name__", type(fn).__name__)
    if isinstance(fn, partial):
        base = fn.func.__name__
        kw = dict(fn.keywords)
        it = int(kw.get("iterations", 1))
        if base == "jacobi":
            return SmootherSpec("jacobi", it, float(np.real(kw.get("omega", 1.0))), name=shown)
        if base == "gauss_seidel":
            return SmootherSpec("gauss_seidel", it, 1.0, kw.get("sweep", "forward"), name=shown)
        if base == "sor":
            return SmootherSpec("sor", it, float(kw["omega"]), kw.get("sweep", "forward"), name=shown)
        if base in ("block_jacobi", "block_gauss_seidel"):
            Dinv = kw.get("Dinv")
            bs = kw.get("blocksize")
            if Dinv is None or bs is None:
                raise NotImplementedError(f"{base} without precomputed Dinv/blocksize")
            Dinv = np.ascontiguousarray(Dinv, dtype=A.dtype)
            if base == "block_jacobi":
                return SmootherSpec("block_jacobi", it, float(np.real(kw.get("omega", 1.0))),
                                    Dinv=Dinv, blocksize=int(bs), name=shown)
            return SmootherSpec("block_gauss_seidel", it, 1.0, kw.get("sweep", "forward"),
                                Dinv=Dinv, blocksize=int(bs), name=shown)
        if base in ("cf_jacobi", "fc_jacobi"):
            if getattr(A, "format", "csr") != "csr":
                raise NotImplementedError(f"{base} on a {A.format} level is not on the device path (CSR only)")
            return SmootherSpec(base, it, float(np.real(kw.get("omega", 1.0))), name=shown,
                                Fpts=np.ascontiguousarray(kw["Fpts"], dtype=np.int32),
                                Cpts=np.ascontiguousarray(kw["Cpts"], dtype=np.int32),
                                f_iterations=int(kw.get("f_iterations", 1)), c_iterations=int(kw.get("c_iterations", 1)))
        if base in ("cf_block_jacobi", "fc_block_jacobi"):
            Dinv, bs = kw.get("Dinv"), kw.get("blocksize")
            if Dinv is None or bs is None or int(bs) < 2:
                raise NotImplementedError(f"{base} without precomputed Dinv / with blocksize 1")
            if getattr(A, "format", "csr") != "bsr" or tuple(A.blocksize) != (int(bs), int(bs)):
                raise NotImplementedError(f"{base}: the level operator must be BSR with {bs}x{bs} blocks on the device path")
            return SmootherSpec(base, it, float(np.real(kw.get("omega", 1.0))), name=shown,
                                Dinv=np.ascontiguousarray(Dinv, dtype=A.dtype), blocksize=int(bs),
                                Fpts=np.ascontiguousarray(kw["Fpts"], dtype=np.int32),
                                Cpts=np.ascontiguousarray(kw["Cpts"], dtype=np.int32),
                                f_iterations=int(kw.get("f_iterations", 1)), c_iterations=int(kw.get("c_iterations", 1)))
        raise NotImplementedError(f"smoother '{base}' is not on the device path")
    cv = _closure_vars(fn)
    if shown in ("gauss_seidel_ne", "gauss_seidel_nr", "jacobi_ne") and "iterations" in cv and "omega" in cv:
        return _normal_equation_spec(shown, A, int(cv["iterations"]), cv.get("sweep", "forward"), cv["omega"])
    if shown == "schwarz" and "subdomain_ptr" in cv and "inv_subblock" in cv:
        # smoothing.py:511-528: the closure holds what schwarz_parameters built at setup; the sweep runs on lvl.Acsr
        return _schwarz_spec(cv["lvl"], A, int(cv["iterations"]), cv["sweep"], cv["subdomain"], cv["subdomain_ptr"],
                             cv["inv_subblock"], cv["inv_subblock_ptr"], shown)
    if shown == "strength_based_schwarz" and "subdomain_ptr" in cv and "lvl" in cv:
        # smoothing.py:531-549: setup_schwarz is called at every application with the subdomains of the strength matrix;
        # the inverted blocks are what schwarz_parameters computes on first use (and caches on lvl.Acsr)
        from .relaxation import schwarz_parameters
        lvl = cv["lvl"]
        Acsr = getattr(lvl, "Acsr", None)
        if Acsr is None:
            Acsr = A.tocsr()
        Acsr.sort_indices()
        sub, sptr, inv, iptr = schwarz_parameters(Acsr, cv["subdomain"], cv["subdomain_ptr"], None, None)
        return _schwarz_spec(lvl, A, int(cv["iterations"]), cv["sweep"], sub, sptr, inv, iptr, shown, Acsr=Acsr)
    if shown in KRYLOV_SMOOTHERS and "tol" in cv and "maxiter" in cv:
        return _krylov_spec(shown, A, cv.get("tol"), cv.get("maxiter"), cv.get("restart"), cv)
    if shown == "none" and not cv:                                  # smoothing.py setup_none: def none(A, x, b): pass
        return SmootherSpec("none", iterations=0, name="None")
    if shown == "chebyshev" and "coefficients" in cv:
        return SmootherSpec("polynomial", int(cv["iterations"]),
                            coefficients=np.asarray(cv["coefficients"], dtype=np.float64).copy(),
                            name="chebyshev")
    if shown == "richardson" and "omega" in cv:
        return SmootherSpec("polynomial", int(cv["iterations"]),
                            coefficients=np.asarray([cv["omega"]], dtype=np.float64),
                            name="richardson")
    raise NotImplementedError(f"smoother '{shown}' is not on the device path")


KRYLOV_SMOOTHERS = ("cg", "gmres", "cgne", "cgnr")


def _krylov_spec(method, A, tol, maxiter, restart, other=None) -> SmootherSpec:
    """cg / gmres / cgne / cgnr as smoother (smoothing.py:794-830) or coarse solver (multilevel.py:752-762).  Only what the
    device loops restate: no preconditioner, callback or residual list inside the cycle, the default stopping criterion,
    GMRES with Householder reflectors."""
    other = other or {}
    for k in ("M", "callback", "residuals"):
        if other.get(k) is not None:
            raise NotImplementedError(f"Krylov smoother '{method}' with {k}= is not on the device path")
    if other.get("criteria", "rr") != "rr" or other.get("orthog", "householder") != "householder":
        raise NotImplementedError(f"Krylov smoother '{method}': only criteria='rr' / orthog='householder' are on the device path")
    if A.dtype.type not in SUPPORTED_DTYPES:
        raise NotImplementedError(f"Krylov smoother '{method}' on a {A.dtype} level is not on the device path")
    if maxiter is not None and int(maxiter) < 1:
        raise ValueError("Number of iterations must be positive")
    At = None
    if method in ("cgne", "cgnr"):
        # the reference applies A.H through SciPy's CSC product: per output the summation order of the sorted CSR rows of A^T
        At = sparse_op(A.T.conj().tocsr())
    return SmootherSpec(method, int(maxiter) if maxiter is not None else 0, name=method, tol=float(tol),
                        restart=int(restart) if restart is not None else 0, At=At)


def _schwarz_spec(lvl, A, iterations, sweep, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr, name, Acsr=None) -> SmootherSpec:
    if A.dtype.type not in SUPPORTED_DTYPES:
        raise NotImplementedError(f"schwarz on a {A.dtype} level is not on the device path")
    if Acsr is None:
        Acsr = getattr(lvl, "Acsr", None)
    if Acsr is None:
        raise NotImplementedError("schwarz smoother without lvl.Acsr")
    # the sweep's operator is lvl.Acsr (sorted rows); ship it separately when the level's own rows are stored differently
    same = A.format == "csr" and A.has_sorted_indices and A.nnz == Acsr.nnz
    Ar = None if same else sparse_op(Acsr)
    return SmootherSpec("schwarz", iterations, 1.0, sweep, name=name, Ar=Ar,
                        subdomain=_as_int32(subdomain, "subdomain"), subdomain_ptr=_as_int32(subdomain_ptr, "subdomain_ptr"),
                        inv_subblock=np.ascontiguousarray(inv_subblock, dtype=A.dtype),
                        inv_subblock_ptr=_as_int32(inv_subblock_ptr, "inv_subblock_ptr"))


def _inv_or_zero(D):
    Dinv = np.zeros_like(D)
    mask = D != 0.0
    Dinv[mask] = 1.0 / D[mask]
    return Dinv


def _normal_equation_spec(kind, A, iterations, sweep, omega) -> SmootherSpec:
    """gauss_seidel_ne / gauss_seidel_nr / jacobi_ne (smoothing.py:641-675): the operands the reference's
    wrappers build on every call -- the inverse squared row / column norms, with the very SciPy expression of
    ``get_diagonal(A, norm_eq=.., inv=True)`` (util/utils.py:583-598), and the column-major form of A."""
    if A.dtype.type is not np.float64:
        # the reference passes a float64 Dinv to a float32 kernel: TypeError (noconvert bindings)
        raise NotImplementedError(f"{kind} on a {A.dtype} level is not on the device path (float64 only)")
    if A.format == "bsr" and tuple(A.blocksize) != (1, 1):
        raise NotImplementedError(f"{kind} on BSR blocks {A.blocksize} is not on the device path")
    n = A.shape[0]
    if kind == "gauss_seidel_nr":
        Mc = A.tocsc()                                  # matrix_asformat(lvl, 'A', 'csc')
        Mc.sort_indices()
        Mt = Mc.T
        D = (Mt.multiply(Mt.conjugate())) @ np.ones((Mt.shape[0],))
        At = SparseOp("csr", (n, n), (1, 1), _as_int32(Mc.indptr, "indptr"), _as_int32(Mc.indices, "indices"),
                      np.ascontiguousarray(Mc.data), "csc")
        # the wrapper forms r = b - A @ x with the CSC matrix (relaxation.py:983): per row that is a sum in
        # ascending column order -- the level's own A gives the same bits only if its rows are sorted
        Ar = None if (A.format in ("csr", "bsr") and A.has_sorted_indices) else sparse_op(Mc.tocsr())
        return SmootherSpec(kind, iterations, float(omega), sweep, Dinv=np.ravel(_inv_or_zero(D)), At=At, Ar=Ar, name=kind)
    M = A.tocsr()
    M.sort_indices()
    D = (M.multiply(M.conjugate())) @ np.ones((M.shape[0],))
    Dinv = np.ravel(_inv_or_zero(D))
    # On a BSR(1,1) level (SA coarse levels) the reference smooths with lvl.Acsr = lvl.A.tocsr(), whose rows come out
    # SORTED, while the cycle's own products keep the level's stored order: ship the sorted copy for the smoother when
    # the two orders differ.  (A CSR level is sorted in place by the reference -- extract() ships it sorted as a whole.)
    Ar = sparse_op(M) if (A.format == "bsr" and not A.has_sorted_indices) else None
    if kind == "gauss_seidel_ne":
        return SmootherSpec(kind, iterations, float(omega), sweep, Dinv=Dinv, Ar=Ar, name=kind)
    Mc = M.tocsc()
    Mc.sort_indices()
    om = A.dtype.type(np.real(omega))                   # type_prep(A.dtype, [omega]); omega2 * Ax[j] in the kernel
    At = SparseOp("csr", (n, n), (1, 1), _as_int32(Mc.indptr, "indptr"), _as_int32(Mc.indices, "indices"),
                  np.ascontiguousarray(om * Mc.data), "csc")
    return SmootherSpec(kind, iterations, float(np.real(omega)), "forward", Dinv=Dinv, At=At, Ar=Ar, name=kind)


# --------------------------------------------------------------------------- coarse solver
_LINEAR_COARSE = ("'pinv'", "'pinv2'", "'lu'", "'cholesky'", "'splu'")
# multilevel.py:765-782: x = 0; setup_<name>(lvl, **kwargs)(A, x, b) with iterations defaulting to 10
_RELAX_COARSE = ("gauss_seidel", "jacobi", "block_gauss_seidel", "schwarz", "block_jacobi", "richardson", "sor", "chebyshev",
                 "jacobi_ne", "gauss_seidel_ne", "gauss_seidel_nr")


def _relaxation_coarse_smoother(ml, cs, A_c):
    """The relaxation method behind ``coarse_solver='gauss_seidel'`` & co. (multilevel.py:765-782): the reference builds
    it from ``smoothing.setup_<name>(lvl, **kwargs)`` at every coarse solve; the keyword arguments live in the closure of
    the solver it returned.  Built once here with the reference's own setup function, so every parameter (a spectral
    radius for Jacobi / Chebyshev / Richardson: cached on the matrix by the reference, hence the same number in its own
    later solves) is the reference's."""
    import importlib
    call = getattr(type(cs), "__call__", None)
    cells = dict(zip(getattr(call.__code__, "co_freevars", ()), [c.cell_contents for c in (call.__closure__ or ())]))
    solve = cells.get("solve")
    inner = _closure_vars(solve) if solve is not None else {}
    if "kwargs" not in inner or "solver" not in inner:
        raise NotImplementedError("coarse solver: cannot read the relaxation method's arguments back")
    smoothing = importlib.import_module(type(ml).__module__.split(".")[0] + ".relaxation.smoothing")
    lvl = type(ml).Level()
    lvl.A = A_c
    return smoother_spec(getattr(smoothing, "setup_" + str(inner["solver"]))(lvl, **dict(inner["kwargs"])), A_c)



_KRYLOV_COARSE = ("cg", "gmres")       # of multilevel.py:752: the two the device Krylov loops restate


_KRYLOV_KWARGS = {"M", "callback", "residuals", "criteria", "orthog"}


def _krylov_coarse_smoother(cs, A_c, method):
    """``coarse_solver='cg' | 'gmres'`` (multilevel.py:752-762): x = fn(A, b, **kwargs)[0] from x0 = None (zeros), with
    tol = set_tol(A.dtype) (util/params.py:28-31) unless the caller gave one."""
    call = getattr(type(cs), "__call__", None)
    cells = dict(zip(getattr(call.__code__, "co_freevars", ()), [c.cell_contents for c in (call.__closure__ or ())]))
    solve = cells.get("solve")
    inner = _closure_vars(solve) if solve is not None else {}
    if "kwargs" not in inner:
        raise NotImplementedError("coarse solver: cannot read the Krylov method's arguments back")
    kw = dict(inner["kwargs"])
    tol = kw.pop("tol", None)
    if tol is None:
        tol = (1e3 * np.finfo(np.single).eps) if A_c.dtype.char.lower() == "f" else (1e6 * np.finfo(np.double).eps)
    maxiter, restart = kw.pop("maxiter", None), kw.pop("restart", None)
    if "restrt" in kw:                                   # gmres's legacy spelling (krylov/_gmres.py: restrt -> restart)
        legacy = kw.pop("restrt")
        if legacy is not None:
            if restart is not None:
                raise ValueError("Only use restart, not restrt (deprecated).")          # krylov/_gmres.py:106-108
            restart = legacy
    if kw.pop("x0", None) is not None:
        raise NotImplementedError("Krylov coarse solver with x0= is not on the device path")
    unknown = set(kw) - _KRYLOV_KWARGS
    if unknown:
        raise NotImplementedError(f"Krylov coarse solver '{method}': arguments {sorted(unknown)} are not on the device path")
    if method == "gmres" and A_c.shape[0] < 2:
        raise NotImplementedError("coarse_solver='gmres' on a 1 x 1 coarsest level is not on the device path (Householder GMRES needs n >= 2)")
    return _krylov_spec(method, A_c, tol, maxiter, restart, kw)


def _coarse_operator(ml, A_c) -> Tuple[str, Optional[np.ndarray], str]:
    cs = ml.coarse_solver
    name = cs.name() if hasattr(cs, "name") else repr(cs)
    if A_c.nnz == 0:                                    # multilevel.py:801-803
        return "zero", None, name
    if name.strip("'") in _RELAX_COARSE:
        return "relax", _relaxation_coarse_smoother(ml, cs, A_c), name
    if name.strip("'") in _KRYLOV_COARSE:
        return "relax", _krylov_coarse_smoother(cs, A_c, name.strip("'")), name
    if name not in _LINEAR_COARSE:
        # the remaining Krylov names of multilevel.py:752 ('bicgstab', 'cgs', 'qmr', 'minres', ...) and callables (:786-788):
        # not linear in b, so nothing to tabulate -- the coarse right-hand side (a handful of values) is handed to the
        # caller's OWN solver object on the host, inside the device cycle (pamg_solver_set_coarse_host)
        if A_c.dtype.type not in SUPPORTED_DTYPES:
            raise NotImplementedError(f"coarse solver {name} on a {A_c.dtype} level is not on the device path")
        return "host", (cs, A_c), name
    n = A_c.shape[0]
    if n > 4096:
        raise NotImplementedError(f"coarsest level too large for a dense device solve (n={n})")
    if name in ("'pinv'", "'pinv2'"):
        # multilevel.py:717-721: P = pinv(A.toarray()) cached on first use, x = np.dot(P, b).
        # Keep the reference's own array (memory order included) -- it IS the operator.
        cs(A_c, np.zeros(n, dtype=A_c.dtype))
        if hasattr(cs, "P"):
            return "dense", np.asarray(cs.P), name
    # the solver is linear in b: tabulate it column by column with the reference's own
    # factorisation (cached inside the GenericSolver on first use, multilevel.py:717-721)
    eye = np.eye(n, dtype=A_c.dtype)
    M = np.empty((n, n), dtype=A_c.dtype)
    for k in range(n):
        M[:, k] = np.ravel(cs(A_c, eye[:, k].copy()))
    return "dense", np.ascontiguousarray(M), name


# --------------------------------------------------------------------------- entry point
def extract(ml) -> HierarchySpec:
    """Read a constructed reference ``MultilevelSolver`` into a HierarchySpec."""
    levels = ml.levels
    if len(levels) == 0:
        raise ValueError("empty hierarchy")
    spec = HierarchySpec()
    for i, lvl in enumerate(levels):
        A = sparse_op(lvl.A)
        if A.shape[0] != A.shape[1]:
            raise ValueError("expected square matrix")
        ls = LevelSpec(A=A)
        if i < len(levels) - 1:
            ls.P = sparse_op(lvl.P)
            ls.R = sparse_op(lvl.R)
            ls.pre = smoother_spec(getattr(lvl, "presmoother", None), lvl.A)
            ls.post = smoother_spec(getattr(lvl, "postsmoother", None), lvl.A)
            kinds = {sm.kind for sm in (ls.pre, ls.post) if sm is not None}
            if kinds & {"gauss_seidel_ne", "jacobi_ne"} and lvl.A.format == "csr" and not lvl.A.has_sorted_indices:
                # the reference's wrappers call get_diagonal(lvl.Acsr) -- lvl.Acsr IS lvl.A for a CSR level -- which
                # sorts the matrix IN PLACE (util/utils.py:583) the first time the smoother runs: from then on
                # every product with this level's A sums in sorted order.  Ship it sorted.
                As = lvl.A.copy()
                As.sort_indices()
                ls.A = sparse_op(As)
        for nm, op in (("A", ls.A), ("P", ls.P), ("R", ls.R)):
            if op is not None and op.dtype != levels[0].A.dtype:
                raise NotImplementedError(
                    f"mixed-precision hierarchy (level {i} {nm} is {op.dtype}, fine level is "
                    f"{levels[0].A.dtype}) is not on the device path")
        spec.levels.append(ls)
    kind, op, spec.coarse_name = _coarse_operator(ml, levels[-1].A)
    spec.coarse_kind = kind
    if kind == "relax":
        spec.coarse_smoother = op
    elif kind == "host":
        spec.coarse_host = op
    else:
        spec.coarse_op = op
    return spec


# --------------------------------------------------------------------------- (de)serialisation
def _put_op(d, key, op: Optional[SparseOp]):
    if op is None:
        return
    d[f"{key}.indptr"], d[f"{key}.indices"], d[f"{key}.data"] = op.indptr, op.indices, op.data
    d[f"{key}.meta"] = np.array([op.shape[0], op.shape[1], op.blocksize[0], op.blocksize[1],
                                 1 if op.fmt == "bsr" else 0], dtype=np.int64)
    d[f"{key}.src"] = np.array(op.src_format)


def _get_op(z, key) -> Optional[SparseOp]:
    if f"{key}.meta" not in z:
        return None
    m = z[f"{key}.meta"]
    return SparseOp("bsr" if m[4] else "csr", (int(m[0]), int(m[1])), (int(m[2]), int(m[3])),
                    z[f"{key}.indptr"], z[f"{key}.indices"], z[f"{key}.data"], str(z[f"{key}.src"]))


def _put_sm(d, key, s: Optional[SmootherSpec]):
    if s is None:
        return
    d[f"{key}.kind"] = np.array(s.kind)
    d[f"{key}.num"] = np.array([s.iterations, s.blocksize], dtype=np.int64)
    d[f"{key}.omega"] = np.array(s.omega, dtype=np.float64)
    d[f"{key}.sweep"] = np.array(s.sweep)
    d[f"{key}.name"] = np.array(s.name)
    if s.coefficients is not None:
        d[f"{key}.coefficients"] = np.asarray(s.coefficients, dtype=np.float64)
    if s.Dinv is not None:
        d[f"{key}.Dinv"] = s.Dinv
    if s.At is not None:
        _put_op(d, f"{key}.At", s.At)
    if s.Ar is not None:
        _put_op(d, f"{key}.Ar", s.Ar)
    if s.subdomain_ptr is not None:
        d[f"{key}.subdomain"], d[f"{key}.subdomain_ptr"] = s.subdomain, s.subdomain_ptr
        d[f"{key}.inv_subblock"], d[f"{key}.inv_subblock_ptr"] = s.inv_subblock, s.inv_subblock_ptr
    if s.kind in KRYLOV_SMOOTHERS:
        d[f"{key}.krylov"] = np.array([s.tol, float(s.restart)], dtype=np.float64)
    if s.Fpts is not None:
        d[f"{key}.Fpts"] = np.asarray(s.Fpts, dtype=np.int32)
        d[f"{key}.Cpts"] = np.asarray(s.Cpts, dtype=np.int32)
        d[f"{key}.fc_iters"] = np.array([s.f_iterations, s.c_iterations], dtype=np.int64)


def _get_sm(z, key) -> Optional[SmootherSpec]:
    if f"{key}.kind" not in z:
        return None
    num = z[f"{key}.num"]
    sm = SmootherSpec(str(z[f"{key}.kind"]), int(num[0]), float(z[f"{key}.omega"]), str(z[f"{key}.sweep"]),
                      z[f"{key}.coefficients"] if f"{key}.coefficients" in z else None,
                      z[f"{key}.Dinv"] if f"{key}.Dinv" in z else None, int(num[1]), str(z[f"{key}.name"]))
    sm.At = _get_op(z, f"{key}.At")
    sm.Ar = _get_op(z, f"{key}.Ar")
    if f"{key}.subdomain_ptr" in z:
        sm.subdomain, sm.subdomain_ptr = z[f"{key}.subdomain"], z[f"{key}.subdomain_ptr"]
        sm.inv_subblock, sm.inv_subblock_ptr = z[f"{key}.inv_subblock"], z[f"{key}.inv_subblock_ptr"]
    if f"{key}.krylov" in z:
        sm.tol, sm.restart = float(z[f"{key}.krylov"][0]), int(z[f"{key}.krylov"][1])
    if f"{key}.Fpts" in z:
        sm.Fpts, sm.Cpts = z[f"{key}.Fpts"], z[f"{key}.Cpts"]
        sm.f_iterations, sm.c_iterations = (int(v) for v in z[f"{key}.fc_iters"])
    return sm


def save_spec(path, spec: HierarchySpec, **extra):
    """Write a HierarchySpec (+ any extra named arrays) to a compressed ``.npz``."""
    if spec.coarse_kind == "host":
        raise NotImplementedError("a hierarchy whose coarse solver runs on the host (a Krylov name other than 'cg' / 'gmres', or a callable) holds a Python "
                                  "object and cannot be written to a file")
    d = {"nlevels": np.array(len(spec.levels)), "coarse_kind": np.array(spec.coarse_kind),
         "coarse_name": np.array(spec.coarse_name)}
    if spec.coarse_op is not None:
        d["coarse_op"] = np.asarray(spec.coarse_op)
        d["coarse_op_fortran"] = np.array(bool(np.isfortran(spec.coarse_op)))
    _put_sm(d, "coarse_smoother", spec.coarse_smoother)
    for i, L in enumerate(spec.levels):
        _put_op(d, f"L{i}.A", L.A)
        _put_op(d, f"L{i}.P", L.P)
        _put_op(d, f"L{i}.R", L.R)
        _put_sm(d, f"L{i}.pre", L.pre)
        _put_sm(d, f"L{i}.post", L.post)
    for k, v in extra.items():
        d[f"extra.{k}"] = np.asarray(v)
    np.savez_compressed(path, **d)


def load_spec(path):
    """Inverse of ``save_spec``: returns (HierarchySpec, dict of extra arrays)."""
    z = np.load(path, allow_pickle=False)
    spec = HierarchySpec(coarse_kind=str(z["coarse_kind"]), coarse_name=str(z["coarse_name"]))
    if "coarse_op" in z:
        M = z["coarse_op"]
        spec.coarse_op = np.asfortranarray(M) if bool(z["coarse_op_fortran"]) else np.ascontiguousarray(M)
    spec.coarse_smoother = _get_sm(z, "coarse_smoother")
    for i in range(int(z["nlevels"])):
        spec.levels.append(LevelSpec(A=_get_op(z, f"L{i}.A"), P=_get_op(z, f"L{i}.P"), R=_get_op(z, f"L{i}.R"),
                                     pre=_get_sm(z, f"L{i}.pre"), post=_get_sm(z, f"L{i}.post")))
    extra = {k[len("extra."):]: z[k] for k in z.files if k.startswith("extra.")}
    return spec, extra
