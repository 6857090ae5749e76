"""Host logic of the persistent Schwarz sweep (CPU, no GPU): the dependency levels and the version table built by pyamg_amd/csrc/pamg_schwarz_plan.h
are replayed by tests/schwarz_emul.cpp the way schwarz_versioned_kernel consumes them -- any number of waves, visited in an adversarial order -- and
must reproduce the oracle's sequential sweep (amg_core::overlapping_schwarz_csr, relaxation.h:1420-1492) BIT FOR BIT, while the replay asserts what the
device relies on: no deadlock for any number of waves, every slot written exactly once, versions of a row in consecutive slots."""
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


@pytest.fixture(scope="module")
def emul():
    out = HERE / "build"
    out.mkdir(exist_ok=True)
    so = out / "schwarz_emul.so"
    src = HERE / "schwarz_emul.cpp"
    hdr = ROOT / "pyamg_amd" / "csrc" / "pamg_schwarz_plan.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.schwarz_emul_sweep_f64.restype = ctypes.c_int
    return lib


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def blocks_of(A, Sp, Sj):
    """inverted diagonal blocks (plain inverses are enough for the replay: both sides get the same arrays)"""
    A = sp.csr_matrix(A)
    Tp = np.zeros(len(Sp), dtype=np.int32)
    Tp[1:] = np.cumsum(np.diff(Sp).astype(np.int64) ** 2)
    Tx = np.zeros(int(Tp[-1]))
    for d in range(len(Sp) - 1):
        rows = Sj[Sp[d]:Sp[d + 1]]
        if rows.size:
            blk = A[rows][:, rows].toarray()
            Tx[Tp[d]:Tp[d + 1]] = np.linalg.pinv(blk).ravel()
    return Tp, Tx


def run(lib, A, Sp, Sj, Tp, Tx, x, b, start, stop, step, waves, max_reads=0):
    A = sp.csr_matrix(A); A.sort_indices()
    Ap, Aj, Ax = i32(A.indptr), i32(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    rc = lib.schwarz_emul_sweep_f64(A.shape[0], p(Ap), p(Aj), p(Ax), len(Sp) - 1, p(i32(Sp)), p(i32(Sj)), p(i32(Tp)), p(Tx), p(xx),
                                    p(np.ascontiguousarray(b, dtype=np.float64)), start, stop, step, waves, ctypes.c_longlong(max_reads), p(stats))
    names = ("visited", "levels", "widest", "slots", "reads", "reads_v0", "rounds", "declined")
    return rc, xx, dict(zip(names, (int(v) for v in stats)))


def reference(A, Sp, Sj, Tp, Tx, x, b, start, stop, step):
    A = sp.csr_matrix(A); A.sort_indices()
    xx = np.array(x, dtype=np.float64)
    orc.overlapping_schwarz_csr(i32(A.indptr), i32(A.indices), np.ascontiguousarray(A.data, dtype=np.float64), xx, np.ascontiguousarray(b, dtype=np.float64),
                                Tx, i32(Tp), i32(Sj), i32(Sp), start, stop, step)
    return xx


@pytest.mark.parametrize("grid", [(23, 19), (9, 8, 7)])
@pytest.mark.parametrize("waves", [1, 7, 64, 100000])
def test_replay_is_the_sequential_sweep_bit_for_bit(emul, grid, waves):
    A = sp.csr_matrix(poisson_csr(grid)); A.sort_indices()
    n = A.shape[0]
    Sp, Sj = A.indptr.copy(), A.indices.copy()                     # the reference's default: subdomain of row i = the pattern of row i
    Tp, Tx = blocks_of(A, Sp, Sj)
    rng = np.random.RandomState(5)
    x, b = rng.rand(n), rng.rand(n)
    for (r0, r1, rs) in ((0, n, 1), (n - 1, -1, -1), (2, 2 + 3 * ((n - 3) // 3), 3)):      # (the reference's loop runs `d != row_stop`: the stride must hit it)
        rc, got, st = run(emul, A, Sp, Sj, Tp, Tx, x, b, r0, r1, rs, waves)
        assert rc == 0, (rc, st)
        want = reference(A, Sp, Sj, Tp, Tx, x, b, r0, r1, rs)
        assert np.array_equal(got, want), (grid, waves, r0, rs)
        assert st["visited"] == len(range(r0, r1, rs)) and st["slots"] == sum(Sp[d + 1] - Sp[d] for d in range(r0, r1, rs))
        assert st["reads_v0"] > 0 and st["levels"] >= 2 * grid[0] // abs(rs) // 2


def test_irregular_subdomains_and_nonsymmetric_pattern(emul):
    rng = np.random.RandomState(9)
    n = 400
    A = sp.random(n, n, density=0.02, random_state=rng, format="csr") + sp.diags(4.0 + rng.rand(n))
    A = sp.csr_matrix(A); A.sort_indices()
    # subdomains of 1 .. 12 distinct rows, some rows in no subdomain, some in many
    Sp, Sj = [0], []
    for _ in range(300):
        size = rng.randint(1, 13)
        Sj += list(rng.choice(n // 2 if rng.rand() < 0.5 else n, size=size, replace=False))
        Sp.append(len(Sj))
    Sp, Sj = np.array(Sp), np.array(Sj)
    Tp, Tx = blocks_of(A, Sp, Sj)
    x, b = rng.rand(n), rng.rand(n)
    nsub = len(Sp) - 1
    for waves in (1, 5, 64):
        for (r0, r1, rs) in ((0, nsub, 1), (nsub - 1, -1, -1)):
            rc, got, st = run(emul, A, Sp, Sj, Tp, Tx, x, b, r0, r1, rs, waves)
            assert rc == 0, (rc, st)
            assert np.array_equal(got, reference(A, Sp, Sj, Tp, Tx, x, b, r0, r1, rs))
    untouched = np.setdiff1d(np.arange(n), Sj)
    assert untouched.size and np.array_equal(got[untouched], x[untouched])


def test_planner_declines_what_the_form_does_not_hold(emul):
    A = sp.csr_matrix(poisson_csr((12, 12))); A.sort_indices()
    n = A.shape[0]
    Sp, Sj = A.indptr.copy(), A.indices.copy()
    Tp, Tx = blocks_of(A, Sp, Sj)
    x, b = np.ones(n), np.ones(n)
    # a subdomain that lists a row twice
    Sj2 = Sj.copy(); Sj2[Sp[5] + 1] = Sj2[Sp[5]]
    rc, _, st = run(emul, A, Sp, Sj2, Tp, Tx, x, b, 0, n, 1, 8)
    assert rc == 2 and st["declined"] == 3
    # the read table beyond its cap
    rc, _, st = run(emul, A, Sp, Sj, Tp, Tx, x, b, 0, n, 1, 8, max_reads=100)
    assert rc == 2 and st["declined"] == 2
    # a row updated more than 255 times
    Sp3 = np.arange(0, 301, dtype=np.int32); Sj3 = np.zeros(300, dtype=np.int32)
    Tp3, Tx3 = blocks_of(A, Sp3, Sj3)
    rc, _, st = run(emul, A, Sp3, Sj3, Tp3, Tx3, x, b, 0, 300, 1, 8)
    assert rc == 2 and st["declined"] == 1
    # bad bounds, empty sweep
    assert run(emul, A, Sp, Sj, Tp, Tx, x, b, 0, n + 1, 1, 8)[0] == 1
    rc, got, st = run(emul, A, Sp, Sj, Tp, Tx, x, b, 3, 3, 1, 8)
    assert rc == 0 and st["visited"] == 0 and np.array_equal(got, x)
