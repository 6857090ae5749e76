"""Host logic of the LINE-SCAN fast-order sweep (CPU, no GPU): the layout built by pyamg_amd/csrc/pamg_line_plan.h is replayed
by tests/line_emul.cpp the way gs_line_kernel consumes it (waves take lines statically, chunk after chunk; per chunk an
inclusive scan of the recurrence x_t = B_t + A_t x_{t-1} over the lanes) and must agree with the oracle's sequential sweep
(amg_core::gauss_seidel / sor_gauss_seidel, relaxation.h:48-76,116-145) to rounding -- 1e-13 relative per sweep -- while the
replay asserts what the device relies on (no deadlock for any number of waves, old operands still old)."""
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
    so = out / "line_emul.so"
    src = HERE / "line_emul.cpp"
    hdr = ROOT / "pyamg_amd" / "csrc" / "pamg_line_plan.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.line_emul_sweep_f64.restype = ctypes.c_int
    return lib


def run_emul(lib, A, x, b, start, stop, step, sor=0, omega=1.0, snapshot=0, waves=7):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.line_emul_sweep_f64(ctypes.c_int(A.shape[0]), p(Ap), p(Aj), p(Ax), p(xx), p(np.ascontiguousarray(b, dtype=np.float64)),
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


@pytest.mark.parametrize("grid", [(300,), (9, 70), (6, 7, 33), (4, 5, 130)])
@pytest.mark.parametrize("waves", [1, 5, 64])
def test_grid_stencils_lines_and_levels(emul, grid, waves):
    A = poisson_csr(grid)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    x, b = rng.random(n), rng.random(n)
    nx = grid[-1] if len(grid) > 1 else grid[0]
    for rng_ in ((0, n, 1), (n - 1, -1, -1), (5, n - 3, 1), (n - 4, 2, -1)):
        rc, got, st = run_emul(emul, A, x, b, *rng_, waves=waves)
        assert rc == 0, (rng_, rc)
        assert close(got, ref_sweep(A, x, b, *rng_)), rng_
        rc, got, _ = run_emul(emul, A, x, b, *rng_, sor=1, omega=1.25, waves=waves)
        assert rc == 0 and close(got, ref_sweep(A, x, b, *rng_, sor=1, omega=1.25))
    rc, _, st = run_emul(emul, A, x, b, 0, n, 1, waves=waves)
    nlines_grid = int(np.prod(grid[:-1])) if len(grid) > 1 else 1
    assert st[2] == nlines_grid                                   # one line per grid line (its chunks of <= 64 rows chained in one wave)
    assert st[3] == (sum(g - 1 for g in grid[:-1]) + 1 if len(grid) > 1 else 1)     # line levels: j + k hyperplanes, not i + j + k
    assert st[0] == 2 * len(grid) - 1                             # entries per row other than the diagonal and the in-line predecessor


def test_duplicate_predecessor_entries_keep_their_slots(emul):
    """Unsummed duplicates are legal CSR (the reference adds every stored entry, relaxation.h:61-68): the FIRST entry of the in-line
    predecessor goes into the recurrence, a second one takes a slot like any other early operand -- the slot count per row has to
    include it (a 1-D chain with duplicated sub-diagonal entries: two slots where the stencil alone needs one)."""
    rng = np.random.default_rng(11)
    n = 3000
    ent = []
    for r in range(n):
        row = [(c, v) for c, v in ((r - 1, -1.0), (r, 2.5), (r + 1, -1.0)) if 0 <= c < n]
        if r % 37 == 3:
            row.append((r - 1, -0.25))
        if r % 13 == 5:
            row.insert(0, (r, 0.5))                                # and a duplicate diagonal: the last stored one wins
        ent.append(row)
    A = sp.csr_array((np.array([v for e in ent for _, v in e]), np.array([c for e in ent for c, _ in e], dtype=np.int32),
                      np.cumsum([0] + [len(e) for e in ent]).astype(np.int32)), shape=(n, n))
    x, b = rng.random(n), rng.random(n)
    for rng_ in ((0, n, 1), (n - 1, -1, -1)):
        rc, got, st = run_emul(emul, A, x, b, *rng_)
        assert rc == 0 and st[0] == 2, (rc, st)
        assert close(got, ref_sweep(A, x, b, *rng_)), rng_


def test_variable_coefficients_zero_diagonals_and_a_nonsymmetric_band(emul):
    rng = np.random.default_rng(5)
    n = 2000
    # banded, variable coefficients, non-symmetric values and pattern (offset +3 only on some rows), zero / missing diagonals
    rows, cols, vals = [], [], []
    for i in range(n):
        for off in (-40, -1, 0, 1, 40):
            j = i + off
            if 0 <= j < n and not (off == 0 and i % 97 == 13):
                rows.append(i); cols.append(j)
                vals.append((6.0 + rng.random()) if off == 0 else -rng.random())
        if i % 5 == 0 and i + 3 < n:
            rows.append(i); cols.append(i + 3); vals.append(-0.3)
    A = sp.csr_array((vals, (rows, cols)), shape=(n, n))
    A = A.tolil(); A[77, 77] = 0.0; A = sp.csr_array(A.tocsr()); A.sort_indices()
    x, b = rng.random(n), rng.random(n)
    rc, got, st = run_emul(emul, A, x, b, 0, n, 1, snapshot=1)
    assert rc == 0, rc
    assert close(got, ref_sweep(A, x, b, 0, n, 1))
    # (backwards the +3 entries are early operands three rows back: chunks of five rows, the planner declines)
    assert run_emul(emul, A, x, b, n - 1, -1, -1, snapshot=1)[0] == 2
    A2 = sp.csr_array(A - sp.diags_array(A.diagonal(3), offsets=3))
    A2.eliminate_zeros()
    for rng_ in ((0, n, 1), (n - 1, -1, -1)):
        rc, got, st = run_emul(emul, A2, x, b, *rng_, snapshot=1, waves=3)
        assert rc == 0, rc
        assert close(got, ref_sweep(A2, x, b, *rng_))
    # an early operand two rows back (offset -2) cuts the chunks to pieces: the planner declines (rc 2), other schedulers keep it
    B = sp.csr_array(sp.diags_array([np.full(n - 2, -1.0), np.full(n - 1, -1.0), np.full(n, 5.0), np.full(n - 1, -1.0)], offsets=[-2, -1, 0, 1]))
    rc, _, _ = run_emul(emul, B, x, b, 0, n, 1)
    assert rc == 2
    # strided sweeps are not this form either
    rc, _, _ = run_emul(emul, poisson_csr((50, 50)), np.zeros(2500), np.ones(2500), 0, 2500, 2)
    assert rc == 2
