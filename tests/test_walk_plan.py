"""Host logic of the LINE-WALK fast-order sweep (CPU, no GPU): the layout built by pyamg_amd/csrc/pamg_walk_plan.h is replayed
by tests/walk_emul.cpp the way gs_walk_kernel consumes it (waves take lines statically and walk them row after row, the
predecessor's new value forwarded inside the wave, operands of other lines polled) and must agree with the oracle's sequential
sweep (amg_core::gauss_seidel / sor_gauss_seidel, relaxation.h:48-76,116-145) to rounding -- 1e-13 relative per sweep -- for
any number of waves, without deadlock."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from tools.problems import poisson_csr

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
TOL = 1e-13


@pytest.fixture(scope="module")
def emul():
    out = HERE / "build"
    out.mkdir(exist_ok=True)
    so = out / "walk_emul.so"
    src = HERE / "walk_emul.cpp"
    hdr = ROOT / "pyamg_amd" / "csrc" / "pamg_walk_plan.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.walk_emul_sweep_f64.restype = ctypes.c_int
    return lib


def run_emul(lib, A, x, b, start, stop, step, sor=0, omega=1.0, snapshot=0, waves=7):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.walk_emul_sweep_f64(ctypes.c_int(A.shape[0]), p(Ap), p(Aj), p(Ax), p(xx), p(np.ascontiguousarray(b, dtype=np.float64)),
                                 start, stop, step, sor, ctypes.c_double(omega), snapshot, waves, p(stats))
    return rc, xx, stats


def ref_sweep(A, x, b, start, stop, step, sor=0, omega=1.0):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    if sor:
        orc.sor_gauss_seidel(Ap, Aj, Ax, xx, b, start, stop, step, omega)
    else:
        orc.gauss_seidel(Ap, Aj, Ax, xx, b, start, stop, step)
    return xx


def close(got, ref):
    return np.max(np.abs(got - ref)) <= TOL * max(1.0, np.max(np.abs(ref)))


def coarse_like(nx, ny, nz, seed):
    """27-point-like operator on a lexicographic grid with random symmetric values + a few longer couplings (offset -2 / +2
    along the line, like the coarse operators of smoothed aggregation), diagonally dominant"""
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-2, -1, 0, 1, 2):
                if dx in (-2, 2) and (dy or dz):
                    continue
                src = idx[max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)]
                dst = idx[max(0, dz):nz - max(0, -dz), max(0, dy):ny - max(0, -dy), max(0, dx):nx - max(0, -dx)]
                rows.append(src.ravel()); cols.append(dst.ravel())
    r, c = np.concatenate(rows), np.concatenate(cols)
    v = -rng.random(r.size)
    S = sp.coo_array((v, (r, c)), shape=(n, n)).tocsr()
    S = S + S.T
    S.setdiag(0)
    S.eliminate_zeros()
    A = sp.csr_array(S + sp.diags_array(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0))
    A.sort_indices()
    return A


@pytest.mark.parametrize("waves", [1, 4, 50])
def test_coarse_like_operator_lines_levels_and_result(emul, waves):
    A = coarse_like(20, 9, 7, 3)
    n = A.shape[0]
    rng = np.random.default_rng(4)
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1), (7, n - 5, 1)):
        rc, got, st = run_emul(emul, A, x, b, *rng_, waves=waves)
        assert rc == 0, (rng_, rc)
        assert close(got, ref_sweep(A, x, b, *rng_)), rng_
        rc, got, _ = run_emul(emul, A, x, b, *rng_, sor=1, omega=0.8, waves=waves)
        assert rc == 0 and close(got, ref_sweep(A, x, b, *rng_, sor=1, omega=0.8))
    rc, _, st = run_emul(emul, A, x, b, 0, n, 1, waves=waves)
    assert st[2] == 9 * 7 and st[5] == n - 9 * 7                   # one line per grid line, every other row forwarded its predecessor
    assert st[3] <= 9 + 2 * 7                                      # line levels ~ j + 2 k, far fewer than the row levels (~ i + 3 j + ...)


def test_stencils_zero_diagonals_strided_and_nonsymmetric(emul):
    rng = np.random.default_rng(6)
    P3 = poisson_csr((6, 7, 40))
    n = P3.shape[0]
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1)):
        rc, got, _ = run_emul(emul, P3, x, b, *rng_)
        assert rc == 0 and close(got, ref_sweep(P3, x, b, *rng_))
    Z = P3.tolil()
    for i in range(3, n, 17):
        Z[i, i] = 0.0
    Z = sp.csr_array(Z.tocsr())
    rc, got, _ = run_emul(emul, Z, x, b, 0, n, 1)
    assert rc == 0 and close(got, ref_sweep(Z, x, b, 0, n, 1))
    # non-symmetric pattern: old values from a snapshot
    N = sp.csr_array(P3 + sp.random_array((n, n), density=0.002, random_state=rng, format="csr"))
    N.sort_indices()
    rc, got, _ = run_emul(emul, N, x, b, 0, n, 1, snapshot=1)
    assert rc == 0 and close(got, ref_sweep(N, x, b, 0, n, 1))
    # a strided sweep visits every other row: no row is coupled to the row visited before it -> declined
    assert run_emul(emul, P3, x, b, 0, n, 2)[0] == 2
    # irregular operator: no lines to speak of -> declined
    R = sp.random_array((500, 500), density=0.02, random_state=rng, format="csr") + sp.diags_array(np.full(500, 9.0))
    assert run_emul(emul, sp.csr_array(R), np.zeros(500), np.ones(500), 0, 500, 1)[0] == 2
