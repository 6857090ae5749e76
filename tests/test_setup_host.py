"""CPU tests of the setup-phase host logic (pyamg_amd/aggregation.py): the restart loop of approximate_spectral_radius
with a NumPy stand-in that has the semantics of the device Arnoldi (pamg_arnoldi_*: all steps are run, the process is
truncated at the first breakdown, the restart vector stays with the process), checked against the reference
(oracle/_ref); the patching of a reference package by device_setup(); and the order-exact sparse product: the host
task plan (csrc/pamg_spg_plan.h) replayed on the CPU by tests/spg_emul.cpp the way the kernels consume it, against
SciPy's `A @ B` -- the same arrays, stored order included.  No device work here."""
import ctypes
import subprocess
import warnings
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

import pyamg_amd.aggregation as ag


class NumpyArnoldi:
    def __init__(self, A, maxiter):
        self.A, self.n = A, A.shape[0]
        self.m = min(self.n, maxiter)
        self.v0, self.planes = None, 0

    def run(self, v0, breakdown):
        if v0 is not None:
            self.v0 = np.ravel(v0).copy()
            self.planes = 2 if np.iscomplexobj(v0) else 1
        V = [self.v0 / np.sqrt(np.vdot(self.v0, self.v0).real)]
        m = self.m
        H = np.zeros((m + 1, m), dtype=complex if self.planes == 2 else float)
        with np.errstate(all="ignore"):
            for j in range(m):
                w = self.A @ V[-1]
                for i, u in enumerate(V):
                    H[i, j] = np.vdot(u, w)
                    w = w - H[i, j] * u
                H[j + 1, j] = np.sqrt(np.vdot(w, w).real)
                V.append(w / H[j + 1, j])
        nc, flag = m, False
        for j in range(m):
            if not (H[j + 1, j].real >= breakdown):
                nc, flag = j + 1, True
                break
        self.V = V
        return H, nc, flag

    def combine(self, coef):
        coef = np.ravel(coef)
        self.v0 = sum(c * v for c, v in zip(coef, self.V[:len(coef)]))
        if np.iscomplexobj(coef):
            self.planes = 2

    def vector(self):
        return self.v0.reshape(-1, 1)

    def free(self):
        pass


def _reference():
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not built")
    import pyamg
    return pyamg


def test_restart_loop_against_the_reference(monkeypatch):
    pyamg = _reference()
    from pyamg.util.linalg import approximate_spectral_radius as ref_rho
    monkeypatch.setattr(ag, "_Arnoldi", NumpyArnoldi)
    rng = np.random.default_rng(3)
    n = 400
    K = sp.diags_array([np.ones(n - 1), -np.ones(n - 1)], offsets=[1, -1], format="csr") * 3.0
    cases = [pyamg.gallery.poisson((30, 30), format="csr"),
             (K + sp.random_array((n, n), density=0.01, random_state=rng, format="csr") * 0.1 + 0.05 * sp.eye_array(n)).tocsr(),   # complex Ritz pair
             sp.csr_array(np.diag([1.0, 2.0, 3.0])),             # maxiter clipped to n, breakdown
             sp.csr_array(np.eye(2))]
    for k, M in enumerate(cases):
        for kw in ({}, {"maxiter": 8, "restart": 2}, {"tol": 1e-6, "maxiter": 20, "restart": 8}):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                np.random.seed(17 + k)
                r_ref = ref_rho(M.copy(), **kw)
                np.random.seed(17 + k)
                v0 = np.random.rand(M.shape[0], 1)
                r = ag._spectral_radius(M, kw.get("tol", 0.01), kw.get("maxiter", 15), kw.get("restart", 5), v0)
            assert abs(r - r_ref) <= 1e-12 * abs(r_ref), (k, kw, r, r_ref)


def test_device_setup_patches_and_restores():
    pyamg = _reference()
    import pyamg.aggregation.aggregation as agg
    import pyamg.relaxation.smoothing as smoothing
    before = (agg.jacobi_prolongation_smoother, agg.richardson_prolongation_smoother, smoothing.approximate_spectral_radius)
    with ag.device_setup(pyamg):
        assert agg.jacobi_prolongation_smoother is ag.jacobi_prolongation_smoother
        assert agg.richardson_prolongation_smoother is ag.richardson_prolongation_smoother
        assert smoothing.approximate_spectral_radius is not before[2]
        # operands the device path does not take stay with the reference function (a LinearOperator's matvec is host code)
        from scipy.sparse.linalg import aslinearoperator
        np.random.seed(0)
        rho = smoothing.approximate_spectral_radius(aslinearoperator(sp.csr_array(np.diag([1.0, 2.0, 3.0]))))
        assert abs(rho - 3.0) < 1e-12
    assert before == (agg.jacobi_prolongation_smoother, agg.richardson_prolongation_smoother, smoothing.approximate_spectral_radius)
    with pytest.raises(RuntimeError):
        with ag.device_setup(pyamg):
            raise RuntimeError("x")
    assert agg.jacobi_prolongation_smoother is before[0]


def test_argument_checks_need_no_device():
    A = sp.csr_array(np.eye(3))
    with pytest.raises(ValueError):
        ag.approximate_spectral_radius(sp.csr_array((3, 4)))
    with pytest.raises(ValueError):
        ag.approximate_spectral_radius(A, maxiter=0)
    with pytest.raises(ValueError):
        ag.approximate_spectral_radius(A, restart=-1)
    with pytest.raises(NotImplementedError):
        ag.approximate_spectral_radius(A.astype(np.complex128))
    A.rho = 42.0
    assert ag.approximate_spectral_radius(A) == 42.0                 # cached value wins, like the reference
    with pytest.raises(NotImplementedError):
        ag.jacobi_prolongation_smoother(sp.csr_array(np.eye(3)), sp.csr_array(np.eye(3)), None, None, filter_entries=True)
    with pytest.raises(ValueError):
        ag.jacobi_prolongation_smoother(sp.csr_array(np.eye(3)), sp.csr_array(np.eye(3)), None, None, weighting="nope")


# --------------------------------------------------------------------------- sparse product: plan + CPU replay
HERE = Path(__file__).resolve().parent


@pytest.fixture(scope="module")
def spg():
    out = HERE / "build"
    out.mkdir(exist_ok=True)
    so = out / "spg_emul.so"
    src = HERE / "spg_emul.cpp"
    hdr = HERE.parent / "pyamg_amd" / "csrc" / "pamg_spg_plan.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def _replay(lib, A, B, col_block=1, keep=0):
    A, B = sp.csr_array(A), sp.csr_array(B)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)       # noqa: E731
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)       # noqa: E731
    Ap, Aj, Bp, Bj = i32(A.indptr), i32(A.indices), i32(B.indptr), i32(B.indices)
    Ax, Bx = np.ascontiguousarray(A.data, dtype=np.float64), np.ascontiguousarray(B.data, dtype=np.float64)
    cap = int((A @ sp.csr_array((np.ones(B.nnz), Bj, Bp), shape=B.shape)).nnz + 16) if A.nnz and B.nnz else 16
    cap = max(cap, int(np.sum(np.diff(Bp)[Aj])) + 16)
    Cp, Cj, Cx = np.zeros(A.shape[0] + 1, dtype=np.int32), np.zeros(cap, dtype=np.int32), np.zeros(cap)
    stats = np.zeros(4, dtype=np.int64)
    rc = lib.spg_emul_f64(A.shape[0], B.shape[1], p(Ap), p(Aj), p(Ax), p(Bp), p(Bj), p(Bx), col_block, keep, p(Cp), p(Cj), p(Cx),
                          ctypes.c_int64(cap), p(stats))
    assert rc == 0, rc
    nnz = Cp[-1]
    return sp.csr_array((Cx[:nnz], Cj[:nnz], Cp), shape=(A.shape[0], B.shape[1])), stats


def _shuffled(M, rng):
    M = sp.csr_array(M).copy()
    for i in range(M.shape[0]):
        lo, hi = M.indptr[i], M.indptr[i + 1]
        q = rng.permutation(hi - lo)
        M.indices[lo:hi], M.data[lo:hi] = M.indices[lo:hi][q], M.data[lo:hi][q]
    M.has_sorted_indices = False
    return M


def _same_arrays(C, ref):
    assert C.nnz == ref.nnz and np.array_equal(C.indptr, ref.indptr)
    assert np.array_equal(C.indices, ref.indices) and np.array_equal(np.ravel(C.data), np.ravel(ref.data))


def test_product_replay_is_scipys_array(spg):
    lim = (ctypes.c_int * 5)()
    spg.spg_emul_limits(lim)
    assert lim[4] * 256 < 2 ** 32                      # a launch slice stays below the 2^32-thread limit
    rng = np.random.default_rng(21)
    for (m, k, n, da, db) in ((300, 200, 250, 0.05, 0.05), (1, 1, 1, 1.0, 1.0), (2000, 1500, 1800, 0.004, 0.006)):
        A = _shuffled(sp.random_array((m, k), density=da, random_state=rng, format="csr"), rng)
        B = _shuffled(sp.random_array((k, n), density=db, random_state=rng, format="csr"), rng)
        C, st = _replay(spg, A, B)
        _same_arrays(C, A @ B)
        assert st[1] == 0
    # exact cancellations are dropped like SciPy drops them
    A = sp.csr_array(np.array([[1.0, -1.0, 0.0], [2.0, 0.0, 1.0], [0.0, 0.0, 0.0]]))
    B = sp.csr_array(np.array([[3.0, 1.0], [3.0, 0.0], [-6.0, 5.0]]))
    C, st = _replay(spg, A, B)
    _same_arrays(C, A @ B)
    assert st[3] == 2


def test_product_replay_long_rows(spg):
    """rows beyond 4096 products: windows of 2048 columns, batches cut by the prefix of in-window products, accumulators
    continued across batches, SciPy's emission order restored over the finished row"""
    rng = np.random.default_rng(22)
    m, k, n = 40, 3000, 9000
    A = sp.random_array((m, k), density=0.002, random_state=rng, format="lil")
    A[3, :] = rng.standard_normal(k)
    A[20, ::2] = rng.standard_normal(k // 2)
    A = _shuffled(A.tocsr(), rng)
    B = _shuffled(sp.random_array((k, n), density=0.003, random_state=rng, format="csr"), rng)
    C, st = _replay(spg, A, B)
    _same_arrays(C, A @ B)
    assert st[1] == 2 and st[2] >= 2 * (n // 2048)      # two long rows, several windows each
    # the shape of a coarse Galerkin product: every row long, few distinct columns, +-1 values (cancellations)
    A2 = _shuffled(sp.random_array((30, 900), density=0.6, random_state=rng, format="csr"), rng)
    A2.data = np.round(A2.data * 4.0)
    A2.eliminate_zeros()
    B2 = sp.random_array((900, 300), density=0.09, random_state=rng, format="csr")
    B2.data = np.sign(B2.data - 0.5)
    C2, st2 = _replay(spg, A2, B2)
    _same_arrays(C2, A2 @ B2)
    assert st2[1] == 30 and st2[3] > 0
    # between 4096 and 8192 products per row: whole-row tasks of their own (the big variant of the expand - sort - compress
    # kernel), no windows -- the rows of the level-1 -> 2 Galerkin products of 3-D problems
    B4 = sp.random_array((900, 300), density=0.04, random_state=rng, format="csr")
    B4.data = np.sign(B4.data - 0.5)
    npr = np.array([(B4.indptr[A2.indices[A2.indptr[i]:A2.indptr[i + 1]] + 1] - B4.indptr[A2.indices[A2.indptr[i]:A2.indptr[i + 1]]]).sum() for i in range(30)])
    assert ((npr > 4096) & (npr <= 8192)).sum() >= 20
    C4, st4 = _replay(spg, A2, B4)
    _same_arrays(C4, A2 @ B4)
    assert st4[1] == int((npr > 8192).sum())
    # a batch that must be cut: B rows of ~1500 in-window entries, 256 candidate entries per batch
    A3 = sp.csr_array(np.ones((2, 40)))
    B3 = sp.random_array((40, 1900), density=0.8, random_state=rng, format="csr")
    C3, st3 = _replay(spg, A3, B3)
    _same_arrays(C3, A3 @ B3)


def test_product_replay_true_blocks(spg):
    """BSR operands with true blocks: the scalar view with whole blocks in FORWARD order of first touch and the zeros inside
    them kept is what SciPy's bsr_matmat stores"""
    rng = np.random.default_rng(23)
    nb, ncb = 60, 12
    pat = sp.random_array((nb, nb), density=0.08, random_state=rng, format="csr")
    Ab = sp.bsr_array(sp.kron(pat + pat.T + 4.0 * sp.eye_array(nb), np.array([[2.0, 0.0], [0.5, 3.0]]), format="bsr"), blocksize=(2, 2))
    Pk = sp.bsr_array(sp.kron(sp.random_array((nb, ncb), density=0.15, random_state=rng, format="csr"), rng.standard_normal((2, 3)), format="bsr"),
                      blocksize=(2, 3))
    ref = Ab @ Pk

    def flat(M):                                         # the device's scalar view: block after block, row-major inside
        R, Cb = M.blocksize
        ip, ix, dat = [0], [], []
        for I in range(M.shape[0] // R):
            for r in range(R):
                for p in range(M.indptr[I], M.indptr[I + 1]):
                    ix.extend(M.indices[p] * Cb + np.arange(Cb))
                    dat.extend(M.data[p, r, :])
                ip.append(len(ix))
        return sp.csr_array((np.array(dat), np.array(ix, dtype=np.int32), np.array(ip, dtype=np.int32)), shape=M.shape)

    C, _ = _replay(spg, flat(Ab), flat(Pk), col_block=3, keep=1)
    Cb = C.tobsr(blocksize=(2, 3))
    assert np.array_equal(Cb.indptr, ref.indptr) and np.array_equal(Cb.indices, ref.indices) and np.array_equal(Cb.data, ref.data)



# --------------------------------------------------------------------------- compressed operator streams (solve phase)
@pytest.fixture(scope="module")
def stream_emul():
    out = HERE / "build"
    out.mkdir(exist_ok=True)
    so = out / "stream_emul.so"
    src = HERE / "stream_emul.cpp"
    hdr = HERE.parent / "pyamg_amd" / "csrc" / "pamg_stream_plan.h"
    hdr2 = HERE.parent / "pyamg_amd" / "csrc" / "pamg_rowmask_map.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime, hdr2.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def _stream_replay(lib, A, x, cap=1536, max_rows=1024):
    A = sp.csr_array(A)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)       # noqa: E731
    Ap, Aj = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    n = A.shape[0]
    ys = [np.zeros(n) for _ in range(3)]
    info = np.zeros(6, dtype=np.int64)
    assert lib.stream_emul_f64(n, p(Ap), p(Aj), p(Ax), p(x), cap, max_rows, p(ys[0]), p(ys[1]), p(ys[2]), p(info)) == 0
    return ys, info


def _rowmask_replay(lib, A, x, kz=4):
    A = sp.csr_array(A)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)       # noqa: E731
    Ap, Aj = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    n = A.shape[0]
    ylin, ylat = np.zeros(n), np.zeros(n)
    info = np.zeros(8, dtype=np.int64)
    assert lib.rowmask_emul_f64(n, p(Ap), p(Aj), p(Ax), p(x), kz, p(ylin), p(ylat), p(info)) == 0
    return ylin, ylat, info


def test_row_masks_deliver_the_csr(stream_emul):
    """plan_row_masks + the row maps of the row-mask kernels (csrc/pamg_rowmask_map.h), replayed on the CPU: a stencil's rows
    as masks over its longest list give back the CSR's own (column, value) pairs in storage order (SciPy's bits), every
    workgroup order visits every row exactly once; lattices that do not fit the 64 x 4 x kz tiles keep the linear form;
    rows that are no sub-list of the longest list walk the CSR arrays; operators whose lists disagree on a value have no
    mask form at all."""
    from tools.problems import poisson_csr
    rng = np.random.default_rng(7)
    # 64 x 32 x 8 lattice: the lattice form applies for kz = 2, 4, 8 (tiles_y = 8: XCD slabs too)
    P3 = poisson_csr((8, 32, 64))                # (nz, ny, nx): rows run fastest along the last extent
    x = rng.random(P3.shape[0])
    ref = P3 @ x
    for kz in (2, 4, 8):
        ylin, ylat, info = _rowmask_replay(stream_emul, P3, x, kz)
        assert info[0] == 7 and info[1] == 0 and info[2] == 1 and info[3] == 0 and info[4] == 0, info
        assert (info[5], info[6]) == (64, 64 * 32) and info[7] == P3.shape[0] // (256 * kz)
        assert np.array_equal(ylin, ref) and np.array_equal(ylat, ref)
    # 24^3: no 64-row tiles -> linear form only; 2-D 5-point: 5 entries, linear form
    for grid, nu in (((24, 24, 24), 7), ((300, 200), 5), ((5000,), 3)):
        A = poisson_csr(grid)
        xa = rng.random(A.shape[0])
        ylin, ylat, info = _rowmask_replay(stream_emul, A, xa)
        assert info[0] == nu and info[2] == 0 and info[3] == 0 and info[4] == 0, (grid, info)
        assert np.array_equal(ylin, A @ xa) and np.isnan(ylat).all()
    # some rows with an extra entry / another value: they are no sub-list of the longest list -> walked through the CSR arrays
    odd = P3.tolil()
    for k in range(50):
        i = 700 + 37 * k
        odd[i, i + 5] = -0.5
        odd[i + 3, i + 2] = -1.25
    odd = sp.csr_array(odd.tocsr())
    ylin, ylat, info = _rowmask_replay(stream_emul, odd, x)
    assert info[0] == 7 and info[1] >= 100 and info[2] == 1 and info[3] == 0 and info[4] == 0, info
    assert np.array_equal(ylin, odd @ x) and np.array_equal(ylat, odd @ x)
    # an anisotropic operator whose boundary rows carry ANOTHER diagonal value: those lists are no sub-lists; too many -> no mask form
    n1 = 4096
    T = sp.diags_array([-np.ones(n1 - 1), 2.0 + (np.arange(n1) % 3 == 0), -np.ones(n1 - 1)], offsets=[-1, 0, 1], format="csr")
    ylin, _, info = _rowmask_replay(stream_emul, T, rng.random(n1))
    assert info[0] == 0 and np.isnan(ylin).all()


def test_compressed_operator_streams_deliver_the_csr(stream_emul):
    """The host plans of the whole-operator kernels' streams (csrc/pamg_stream_plan.h: column windows, value codes, row
    patterns) decoded on the CPU the way the kernels decode them: every form hands back the CSR's own (column, value)
    pairs in storage order -- so y = A x has SciPy's bits -- on a 3-D stencil (27 lists), a stencil with more odd rows than
    the table holds (irregular rows walk the code arrays), operators with 256 / 257 distinct values, unsorted rows, and a
    wide operator whose ranges need a fifth column window."""
    from tools.problems import poisson_csr
    rng = np.random.default_rng(4)
    P3 = poisson_csr((24, 24, 24))
    x = rng.random(P3.shape[0])
    (y16, y8, ypat), info = _stream_replay(stream_emul, P3, x)
    ref = P3 @ x
    assert info[0] == 1 and info[1] == 2 and info[2] == 27 and info[3] == 0 and info[4] == 0
    assert np.array_equal(y16, ref) and np.array_equal(y8, ref) and np.array_equal(ypat, ref)
    # range plans of the row-gather / row-pattern kernels (512 rows) give the same
    (_, _, ypat2), info2 = _stream_replay(stream_emul, P3, x, cap=3584, max_rows=512)
    assert np.array_equal(ypat2, ref) and info2[5] < info[5]
    # more different rows than the table holds
    odd = P3.tolil()
    extra = 1.0 + np.arange(20) / 32.0
    for k in range(400):
        i = 3000 + 23 * k
        odd[i, i - 1] = -extra[k % 20]
        odd[i, i + 1] = -extra[k // 20]
    odd = sp.csr_array(odd.tocsr())
    (y16, y8, ypat), info = _stream_replay(stream_emul, odd, x)
    assert info[1] == 21 and info[2] == 245 and info[3] > 100 and info[4] == 0       # 245 lists of 8 entries fill the 24 KB table
    assert np.array_equal(ypat, odd @ x) and np.array_equal(y8, odd @ x)
    # 256 distinct values (codes) and 257 (none); unsorted rows keep their order
    nn = 6000
    cols = (np.arange(nn)[:, None] + np.array([-40, -1, 0, 1, 40])[None, :]) % nn
    v256 = np.concatenate([[0.0, -0.0], rng.standard_normal(254)])
    for vals, nv in ((v256, 256), (np.concatenate([v256, [7.25]]), 0)):
        data = rng.choice(vals, size=cols.size)
        data[:vals.size] = vals
        B = sp.csr_array((data, cols.ravel().astype(np.int32), np.arange(0, cols.size + 1, 5, dtype=np.int32)), shape=(nn, nn))
        xb = rng.random(nn)
        (y16, y8, ypat), info = _stream_replay(stream_emul, B, xb)
        assert info[0] == 1 and info[1] == nv and info[4] == 0
        assert np.array_equal(y16, B @ xb)
        assert np.array_equal(y8, B @ xb) if nv else np.isnan(y8).all()
    # a wide random operator: some range needs a fifth window of 16 K columns -> no 16-bit stream at all
    nw = 90000
    W = sp.csr_array((rng.random(nw * 12), rng.integers(0, nw, size=nw * 12).astype(np.int32), np.arange(0, nw * 12 + 1, 12, dtype=np.int32)),
                     shape=(nw, nw))
    (y16, _, _), info = _stream_replay(stream_emul, W, rng.random(nw))
    assert info[0] == 0 and np.isnan(y16).all()
