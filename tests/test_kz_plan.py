"""Host logic of the lane-parallel FAST-ORDER Kaczmarz sweeps (CPU, no GPU): the layout built by pyamg_amd/csrc/pamg_kz_plan.h (lines by
dependency level over shared indices, the version every entry must see) is replayed by tests/kz_emul.cpp the way kz_lane_kernel consumes
it -- versioned slots, K products per lane, XOR butterfly, the reference's step -- with the waves visited in the adversarial order, and
must agree with the reference's sequential loops (amg_core::gauss_seidel_ne / gauss_seidel_nr, relaxation.h:875-904, 939-975; here: the
oracle's restatement) to rounding: 1e-13 relative per sweep."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
TOL = 1e-13


@pytest.fixture(scope="module")
def emul():
    out = HERE / "build"
    out.mkdir(exist_ok=True)
    so = out / "kz_emul.so"
    src = HERE / "kz_emul.cpp"
    hdrs = [ROOT / "pyamg_amd" / "csrc" / h for h in ("pamg_kz_plan.h", "pamg_lane_plan.h")]
    if not so.exists() or so.stat().st_mtime < max([src.stat().st_mtime] + [h.stat().st_mtime for h in hdrs]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.kz_emul_sweep_f64.restype = ctypes.c_int
    return lib


def seq_ne(A, x, b, Dinv, omega, rows):
    """amg_core::gauss_seidel_ne, relaxation.h:889-902"""
    x = x.copy()
    Ap, Aj, Ax = A.indptr, A.indices, A.data
    for i in rows:
        d = 0.0
        for p in range(Ap[i], Ap[i + 1]):
            d += Ax[p] * x[Aj[p]]
        d = (b[i] - d) * Dinv[i] * omega
        for p in range(Ap[i], Ap[i + 1]):
            x[Aj[p]] += Ax[p] * d
    return x


def seq_nr(At, x, z, Dinv, omega, cols):
    """amg_core::gauss_seidel_nr, relaxation.h:954-973 (At = CSR of A^T = CSC of A)"""
    x, z = x.copy(), z.copy()
    Ap, Aj, Ax = At.indptr, At.indices, At.data
    for i in cols:
        d = 0.0
        for p in range(Ap[i], Ap[i + 1]):
            d += Ax[p] * z[Aj[p]]
        d *= Dinv[i] * omega
        x[i] += d
        for p in range(Ap[i], Ap[i + 1]):
            z[Aj[p]] -= d * Ax[p]
    return x, z


def run(lib, L, v, b, Dinv, omega, nr, xout, start, stop, step, waves):
    L = sp.csr_array(L)
    Lp = np.ascontiguousarray(L.indptr, dtype=np.int32)
    Lj = np.ascontiguousarray(L.indices, dtype=np.int32)
    Lx = np.ascontiguousarray(L.data, dtype=np.float64)
    vv, xx = np.array(v, dtype=np.float64), np.array(xout, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.kz_emul_sweep_f64(L.shape[0], L.shape[1], p(Lp), p(Lj), p(Lx), p(vv), p(np.ascontiguousarray(b, dtype=np.float64)),
                               p(np.ascontiguousarray(Dinv, dtype=np.float64)), ctypes.c_double(omega), nr, p(xx), start, stop, step, waves, p(stats))
    return rc, vv, xx, stats


def close(a, b):
    return np.max(np.abs(a - b)) <= TOL * max(1.0, np.max(np.abs(b)))


def operators():
    rng = np.random.default_rng(2)
    n = 900
    conv = sp.diags_array([np.full(n, 4.0), -np.ones(n - 1), -2 * np.ones(n - 30), -np.ones(n - 30)], offsets=[0, -1, -30, 30], shape=(n, n)).tocsr()
    dense_rows = sp.random_array((400, 400), density=0.12, random_state=rng, format="csr") + sp.diags_array(np.full(400, 9.0))
    rect = sp.random_array((300, 500), density=0.03, random_state=rng, format="csr")
    return [sp.csr_array(conv), sp.csr_array(dense_rows), sp.csr_array(rect)]


@pytest.mark.parametrize("waves", [1, 5, 64, 1000])
def test_ne_and_nr_sweeps_agree_with_the_sequential_loops(emul, waves):
    rng = np.random.default_rng(7)
    for A in operators():
        A.sort_indices()
        m, n = A.shape
        # gauss_seidel_ne: lines = rows of A, v = x
        x, b = rng.random(n), rng.random(m)
        Dinv = 1.0 / np.asarray(A.multiply(A).sum(axis=1)).ravel().clip(1e-30)
        for rng_ in ((0, m, 1), (m - 1, -1, -1), (5, m - 5, 1)):
            rc, got, _, st = run(emul, A, x, b, Dinv, 0.9, 0, np.zeros(m), *rng_, waves)
            assert rc == 0, (rng_, rc)
            assert close(got, seq_ne(A, x, b, Dinv, 0.9, range(*rng_)))
        # gauss_seidel_nr: lines = columns of A (rows of A^T), v = the running residual, xout = x
        At = sp.csr_array(A.T.tocsr())
        At.sort_indices()
        xs, z = rng.random(n), rng.random(m)
        Dn = 1.0 / np.asarray(At.multiply(At).sum(axis=1)).ravel().clip(1e-30)
        for rng_ in ((0, n, 1), (n - 1, -1, -1)):
            rc, zgot, xgot, st = run(emul, At, z, np.zeros(n), Dn, 1.1, 1, xs, *rng_, waves)
            assert rc == 0, (rng_, rc)
            xr, zr = seq_nr(At, xs, z, Dn, 1.1, range(*rng_))
            assert close(xgot, xr) and close(zgot, zr)
            assert st[5] >= 1 and st[4] >= 1


def test_a_line_holding_an_index_twice_is_declined(emul):
    A = sp.csr_array((np.array([1.0, 2.0, 3.0]), np.array([0, 0, 1], dtype=np.int32), np.array([0, 2, 3], dtype=np.int32)), shape=(2, 2))
    rc, *_ = run(emul, A, np.ones(2), np.ones(2), np.ones(2), 1.0, 0, np.zeros(2), 0, 2, 1, 1)
    assert rc == 2
