"""Host logic of the WORKGROUP-RESIDENT fast-order sweep (CPU, no GPU): the layout built by pyamg_amd/csrc/pamg_wg_plan.h is replayed by
tests/wg_emul.cpp the way gs_wg_kernel consumes it -- every tile keeps its rows' values in an array of its own (the LDS copy), walks
its rounds level after level, K products per lane, XOR butterfly over the lanes of a row, (b - sum) * (1 / a_ii); values of other tiles
through the sentinel hand-off -- and must agree with the oracle's sequential sweep (amg_core::gauss_seidel / sor_gauss_seidel,
relaxation.h:48-76,116-145) to rounding, 1e-13 relative per sweep, while the replay asserts what the device relies on: the barrier between
the dependency levels of a tile orders every in-tile operand, tiles only wait for earlier tiles (no deadlock in the adversarial order),
old values outside a tile are still old, every row is done once."""
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
    so = out / "wg_emul.so"
    src = HERE / "wg_emul.cpp"
    hdrs = [ROOT / "pyamg_amd" / "csrc" / h for h in ("pamg_wg_plan.h", "pamg_lane_plan.h", "pamg_tile_plan.h")]
    if not so.exists() or so.stat().st_mtime < max([src.stat().st_mtime] + [h.stat().st_mtime for h in hdrs]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.wg_emul_sweep_f64.restype = ctypes.c_int
    return lib


def run_emul(lib, A, x, b, start, stop, step, max_tile_rows=18432, tiles=0, sor=0, omega=1.0, snapshot=0):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.wg_emul_sweep_f64(ctypes.c_int(A.shape[0]), p(Ap), p(Aj), p(Ax), p(xx), p(np.ascontiguousarray(b, dtype=np.float64)),
                               start, stop, step, max_tile_rows, tiles, sor, ctypes.c_double(omega), snapshot, p(stats))
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


def sa_like(n, per_row, seed, band=40):
    rng = np.random.default_rng(seed)
    i = np.repeat(np.arange(n), per_row // 2)
    j = np.clip(i + rng.integers(-band, band + 1, size=i.size), 0, n - 1)
    v = -rng.random(i.size)
    S = sp.coo_array((v, (i, j)), shape=(n, n)).tocsr()
    S = S + S.T
    S.setdiag(0)
    S.eliminate_zeros()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0
    A = (S + sp.diags_array(d)).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("tiles", [0, 2, 3, 7])
@pytest.mark.parametrize("per_row", [12, 30, 70, 150])
def test_sa_like_rows_any_number_of_tiles(emul, per_row, tiles):
    A = sa_like(3000, per_row, per_row)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1), (100, n - 100, 1)):
        rc, got, st = run_emul(emul, A, x, b, *rng_, tiles=tiles)
        assert rc == 0, (rng_, rc)
        assert st[2] == max(1, tiles)
        if st[2] == 1:
            assert st[5] == 0                                   # one tile: nothing is polled
        else:
            assert st[5] > 0 and st[7] > 0                      # new values cross tile boundaries and are published
        assert close(got, ref_sweep(A, x, b, *rng_))
        rc, got, _ = run_emul(emul, A, x, b, *rng_, tiles=tiles, sor=1, omega=1.3)
        assert rc == 0 and close(got, ref_sweep(A, x, b, *rng_, sor=1, omega=1.3))


def test_small_tiles_force_many_tiles_and_too_many_are_declined(emul):
    A = sa_like(5000, 30, 3)
    n = A.shape[0]
    rng = np.random.default_rng(6)
    x, b = rng.random(n), rng.random(n)
    rc, got, st = run_emul(emul, A, x, b, 0, n, 1, max_tile_rows=1000)          # 5 tiles of 1000 rows
    assert rc == 0 and st[2] == 5 and close(got, ref_sweep(A, x, b, 0, n, 1))
    rc, _, _ = run_emul(emul, A, x, b, 0, n, 1, max_tile_rows=500)              # 10 tiles: more than the form takes
    assert rc == 2


@pytest.mark.parametrize("grid", [(40,), (17, 13), (9, 8, 7)])
def test_stencils_and_partial_sweeps(emul, grid):
    A = poisson_csr(grid)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1), (3, n - 4, 2), (n - 2, 0, -3)):
        if (rng_[1] - rng_[0]) % rng_[2]:
            continue
        for tiles in (0, 2):
            rc, got, st = run_emul(emul, A, x, b, *rng_, tiles=tiles)
            assert rc == 0, (rng_, rc)
            assert close(got, ref_sweep(A, x, b, *rng_))


def test_zero_missing_diagonals_and_nonsymmetric_patterns(emul):
    A = sa_like(300, 12, 1).tolil()
    A[5, 5] = 0.0
    A = A.tocsr()
    C = A.tocoo()
    keep = ~((C.row == 9) & (C.col == 9)) & (C.row != 11)
    A = sp.csr_array((C.data[keep], (C.row[keep], C.col[keep])), shape=A.shape)
    A.sort_indices()
    n = A.shape[0]
    rng = np.random.default_rng(7)
    x, b = rng.random(n), rng.random(n)
    for tiles in (0, 3):
        rc, got, _ = run_emul(emul, A, x, b, 0, n, 1, tiles=tiles, snapshot=1)
        assert rc == 0 and close(got, ref_sweep(A, x, b, 0, n, 1))
    N = sp.random_array((400, 400), density=0.03, random_state=np.random.default_rng(11), format="csr") + sp.diags_array(np.full(400, 8.0))
    N = sp.csr_array(N)
    N.sort_indices()
    x, b = rng.random(400), rng.random(400)
    for tiles in (0, 2, 4):
        for rng_ in ((0, 400, 1), (399, -1, -1)):
            rc, got, _ = run_emul(emul, N, x, b, *rng_, tiles=tiles, snapshot=1)
            assert rc == 0 and close(got, ref_sweep(N, x, b, *rng_))


def test_rows_too_long_are_declined(emul):
    n = 600
    A = sp.csr_array(np.ones((n, n)) + np.diag(np.full(n, n * 2.0)))
    rc, _, _ = run_emul(emul, A, np.zeros(n), np.ones(n), 0, n, 1)
    assert rc == 2
