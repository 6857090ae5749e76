"""Host logic of the MERGED lane-parallel Gauss-Seidel sweep (CPU, no GPU): the plan built by pyamg_amd/csrc/pamg_lanem_plan.h -- s consecutive
dependency levels of the reference's sweep eliminated algebraically into one super-level -- is replayed by tests/lanem_emul.cpp the way
gs_lanem_kernel consumes it and must agree with the oracle's sequential sweep (amg_core::gauss_seidel, relaxation.h:48-76) to rounding
(1e-13 relative per sweep), while the replay asserts what the device relies on: polled operands come from EARLIER super-levels and smaller
group numbers (deadlock freedom for any number of waves, visited in the adversarial order), gates are ancestors, padding carries no product,
every visited row is written exactly once, rows without a usable diagonal stay untouched."""
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
    so = out / "lanem_emul.so"
    src = HERE / "lanem_emul.cpp"
    hdrs = [ROOT / "pyamg_amd" / "csrc" / h for h in ("pamg_lanem_plan.h", "pamg_lane_plan.h", "pamg_tile_plan.h")]
    if not so.exists() or so.stat().st_mtime < max([src.stat().st_mtime] + [h.stat().st_mtime for h in hdrs]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.lanem_emul_sweep_f64.restype = ctypes.c_int
    return lib


def run_emul(lib, A, x, b, start, stop, step, s_max, growth_cap=1e3, len_cap=512, waves=0, plan_only=0, rpw=1):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(16, dtype=np.int64)
    gs = np.zeros(2)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.lanem_emul_sweep_f64(ctypes.c_int(A.shape[0]), p(Ap), p(Aj), p(Ax), p(xx), p(np.ascontiguousarray(b, dtype=np.float64)),
                                  start, stop, step, s_max, ctypes.c_double(growth_cap), len_cap, p(stats), p(gs), waves, plan_only, rpw)
    names = ("super", "levels", "rows", "units", "early", "old", "b", "direct", "max_len", "closed_len", "closed_growth", "widest", "k1", "k2", "k3", "k4")
    return rc, xx, dict(zip(names, (int(v) for v in stats)), growth=float(gs[0]))


def ref_sweep(A, x, b, start, stop, step):
    A = sp.csr_array(A)
    xx = np.array(x, dtype=np.float64)
    orc.gauss_seidel(np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32),
                     np.ascontiguousarray(A.data, dtype=np.float64), xx, b, start, stop, step)
    return xx


def sa_like(n=3000, density=0.006, seed=5):
    rng = np.random.RandomState(seed)
    S = sp.random(n, n, density=density, random_state=rng, format="csr")
    S = sp.csr_array(-abs(S + S.T))
    S.setdiag(0)
    S.eliminate_zeros()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 0.5 + rng.rand(n)
    A = sp.csr_array(S + sp.diags_array(d))
    A.sort_indices()
    return A


def test_merged_sweep_agrees_with_the_sequential_sweep(emul):
    rng = np.random.RandomState(0)
    cases = {"sa_like": sa_like(), "stencil27": None, "poisson3d": poisson_csr((14, 12, 13)), "nonsym": None}
    g = np.arange(12 ** 3).reshape(12, 12, 12)
    import itertools
    rows, cols = [], []
    for dx, dy, dz in itertools.product((-1, 0, 1), repeat=3):
        src = g[max(0, -dx):12 - max(0, dx), max(0, -dy):12 - max(0, dy), max(0, -dz):12 - max(0, dz)]
        dst = g[max(0, dx):12 - max(0, -dx), max(0, dy):12 - max(0, -dy), max(0, dz):12 - max(0, -dz)]
        rows.append(src.ravel()); cols.append(dst.ravel())
    r, c = np.concatenate(rows), np.concatenate(cols)
    cases["stencil27"] = sp.csr_array((np.where(r == c, 26.5, -1.0), (r, c)), shape=(12 ** 3, 12 ** 3))
    N = sp.random(2500, 2500, density=0.005, random_state=rng, format="csr")
    cases["nonsym"] = sp.csr_array(N + sp.diags_array(rng.rand(2500) + 6.0))          # structurally non-symmetric: late rows read old values of rows they are not adjacent to
    for name, A in cases.items():
        n = A.shape[0]
        x, b = rng.rand(n), rng.rand(n)
        for (start, stop, step) in ((0, n, 1), (n - 1, -1, -1), (5, n - 7, 1)):
            ref = ref_sweep(A, x, b, start, stop, step)
            supers = []
            for s in (1, 2, 3, 5, 8):
                for waves in (0, 7, 64):
                    rc, got, st = run_emul(emul, A, x, b, start, stop, step, s, waves=waves)
                    assert rc == 0, (name, s, waves, rc)
                    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref)), (name, s, waves, np.max(np.abs(got - ref)))
                    # two rows per wave (32 lanes each, rows of a super-level paired by length): the same rows, the same super-levels
                    rc2, got2, st2 = run_emul(emul, A, x, b, start, stop, step, s, waves=waves, rpw=2)
                    if rc2 == 2:
                        assert st["max_len"] > 256, (name, s)                      # only rows beyond 8 x 32 operands make the pair form decline
                        continue
                    assert rc2 == 0, (name, s, waves, rc2)
                    assert np.max(np.abs(got2 - ref)) <= TOL * np.max(np.abs(ref)), (name, s, waves)
                    assert st2["rows"] == st["rows"] and st2["early"] + st2["old"] + st2["b"] > 0 or st["direct"] == 0
                supers.append(st["super"])
                assert st["rows"] == len(range(start, stop, step)) and st["b"] >= 0
                if s == 1:
                    assert st["super"] == st["levels"] and st["b"] == 0 and st["early"] + st["old"] == st["direct"]
            assert supers[1] <= (supers[0] + 1) // 2 + st["closed_len"] + st["closed_growth"], (name, supers)     # s = 2 halves the hand-offs
            assert supers[-1] < supers[0]


def test_rows_without_a_diagonal_stay_untouched(emul):
    """relaxation.h:72-74: a row with a zero / missing diagonal is skipped -- as an in-group operand its NEW value is its OLD value"""
    rng = np.random.RandomState(3)
    A = sp.lil_array(sa_like(1500, 0.01, seed=9))
    for i in range(0, 1500, 5):
        A[i, i] = 0.0
    A = sp.csr_array(A)
    Z = sp.csr_array(A)
    Z.eliminate_zeros()                                   # missing instead of explicit zero
    for M in (A, Z):
        n = M.shape[0]
        x, b = rng.rand(n), rng.rand(n)
        for (start, stop, step) in ((0, n, 1), (n - 1, -1, -1)):
            ref = ref_sweep(M, x, b, start, stop, step)
            for s in (2, 4):
                rc, got, st = run_emul(emul, M, x, b, start, stop, step, s, waves=16)
                assert rc == 0 and np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref)), (s, rc)
                assert np.array_equal(got[0::5], x[0::5])


def test_growth_bound_closes_groups(emul):
    """an operator that is NOT diagonally dominant: |a_ir / a_rr| > 1 along chains, the eliminated rows' coefficients grow -- the planner
    closes the super-level in front of the level whose growth factor exceeds the cap (down to s = 1: the unmerged row), and whatever it
    keeps still reproduces the sequential sweep"""
    n = 400
    rng = np.random.RandomState(1)
    main, off = np.full(n, 1.0), np.full(n - 1, -3.0)
    off[23::24] = 0.0                                                  # chains of 24 rows: the sweep itself stays finite (3^23), an 8-level group would not stay below the cap (3^7)
    A = sp.csr_array(sp.diags_array([off, main, 0.1 * off], offsets=[-1, 0, 1]))
    x, b = rng.rand(n), rng.rand(n)
    ref = ref_sweep(A, x, b, 0, n, 1)
    rc, got, st = run_emul(emul, A, x, b, 0, n, 1, 8, growth_cap=1e3)
    assert rc == 0 and st["closed_growth"] > 0 and st["growth"] <= 1e3, st
    assert st["super"] > (st["levels"] + 7) // 8                      # fewer levels per group than asked for
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref))     # growth <= 1e3 costs at most three digits
    rc, got1, st1 = run_emul(emul, A, x, b, 0, n, 1, 8, growth_cap=1.0)   # cap 1: nothing may be merged
    assert rc == 0 and st1["super"] == st1["levels"] and st1["b"] == 0
    assert np.max(np.abs(got1 - ref)) <= TOL * np.max(np.abs(ref))


def test_length_cap_and_unfit_rows(emul):
    rng = np.random.RandomState(2)
    D = sp.random(600, 600, density=0.15, random_state=rng, format="csr")      # ~180 entries per row: merged rows hit the 256-operand cap at once
    D = sp.csr_array(D + D.T + sp.diags_array(rng.rand(600) + 200.0))
    n = D.shape[0]
    x, b = rng.rand(n), rng.rand(n)
    ref = ref_sweep(D, x, b, 0, n, 1)
    rc, got, st = run_emul(emul, D, x, b, 0, n, 1, 4, len_cap=256)
    assert rc == 0 and st["max_len"] <= 256 and st["closed_len"] > 0, st
    assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))
    W = sp.random(1300, 1300, density=0.5, random_state=rng, format="csr")     # rows beyond 512 operands even unmerged: the form declines
    W = sp.csr_array(W + sp.diags_array(np.full(1300, 900.0)))
    rc, _, _ = run_emul(emul, W, rng.rand(1300), rng.rand(1300), 0, 1300, 1, 2, plan_only=1)
    assert rc == 2
    rc, got, st = run_emul(emul, sa_like(800, 0.02), rng.rand(800), rng.rand(800), 0, 800, 1, 3, len_cap=64)
    assert rc == 0 and st["max_len"] <= 64 and st["k2"] == st["k3"] == st["k4"] == 0
