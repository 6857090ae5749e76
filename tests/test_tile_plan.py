"""Host logic of the TILED order-exact sweep (CPU, no GPU): the plan built by
pyamg_amd/csrc/pamg_tile_plan.h is replayed by tests/tile_plan_emul.cpp the way gs_tile_kernel consumes it
(packed step blocks, LDS ring with wrap-around, global hand-off with sentinel, publish flags, OLD operands fetched
several steps ahead)
under three interleavings of the tiles, and must reproduce the oracle's sequential sweep
(amg_core::gauss_seidel / sor_gauss_seidel / bsr_gauss_seidel, relaxation.h:48-76,116-145,185-266) bit for bit."""
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
    so = out / "tile_plan_emul.so"
    src = HERE / "tile_plan_emul.cpp"
    hdr = ROOT / "pyamg_amd" / "csrc" / "pamg_tile_plan.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.tile_emul_sweep_f64.restype = ctypes.c_int
    return lib


def run_emul(lib, A, x, b, start, stop, step, G, W, cap, max_rows, epi=0, omega=1.0, snapshot=0, policy=0, look=5, partition=0):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.tile_emul_sweep_f64(ctypes.c_int(A.shape[0]), p(Ap), p(Aj), p(Ax), p(xx), p(np.ascontiguousarray(b)), start, stop, step,
                                 G, W, cap, min(max_rows, 64), epi, ctypes.c_double(omega), snapshot, policy, look, partition, p(stats))
    assert rc == 0
    return xx, stats


def ref_sweep(A, x, b, start, stop, step, epi=0, omega=1.0):
    A = sp.csr_array(A)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int32)
    Aj = np.ascontiguousarray(A.indices, dtype=np.int32)
    Ax = np.ascontiguousarray(A.data, dtype=np.float64)
    xx = np.array(x, dtype=np.float64)
    if epi == 0:
        orc.gauss_seidel(Ap, Aj, Ax, xx, b, start, stop, step)
    elif epi == 1:
        orc.bsr_gauss_seidel(Ap, Aj, Ax, xx, b, start, stop, step, 1)
    else:
        orc.sor_gauss_seidel(Ap, Aj, Ax, xx, b, start, stop, step, omega)
    return xx


def sym_random(n, density, seed):
    R = sp.random(n, n, density=density, format="csr", random_state=seed)
    A = sp.csr_array(R + R.T + sp.eye_array(n) * 4.0)
    A.sort_indices()
    return A


def operators():
    ops = {"poisson2d": poisson_csr((23, 17)), "poisson3d": poisson_csr((9, 8, 7)), "symrand": sym_random(400, 0.03, 3)}
    N = sp.csr_array(sp.random(300, 300, density=0.04, format="csr", random_state=5) + sp.eye_array(300) * 3.0)
    N.sort_indices()
    ops["nonsym"] = N
    Z = sym_random(200, 0.05, 7).tolil()
    Z[5, 5] = 0.0                      # stored zero diagonal
    Z = sp.csr_array(Z.tocsr())
    Z.sort_indices()
    ops["zerodiag"] = Z
    M = sym_random(150, 0.06, 9).tolil()
    M[11, 11] = 0.0
    M = sp.csr_array(M.tocsr())
    M.eliminate_zeros()                # missing diagonal
    ops["missingdiag"] = M
    return ops


OPS = operators()


@pytest.mark.parametrize("name", sorted(OPS))
@pytest.mark.parametrize("policy", [0, 1, 2])
def test_replay_is_bit_exact(emul, name, policy):
    A = OPS[name]
    n = A.shape[0]
    rng = np.random.RandomState(1)
    x = rng.rand(n)
    b = rng.rand(n)
    snapshot = 1 if name == "nonsym" else 0
    for (start, stop, step) in [(0, n, 1), (n - 1, -1, -1), (3, n - 2, 2), (n - 4, 1, -3)]:
        span = stop - start
        if span % step:
            stop = start + (span // step) * step
        ref = ref_sweep(A, x, b, start, stop, step)
        for (G, W, cap, mr) in [(1, 64, 1020, 64), (3, 64, 40, 8), (7, 256, 508, 64), (16, 64, 16, 4), (64, 128, 100, 64)]:
            got, st = run_emul(emul, A, x, b, start, stop, step, G, W, cap, mr, snapshot=snapshot, policy=policy)
            assert st[6] == 0 and st[7] == 0, (name, G, W, cap, st)
            assert np.array_equal(got, ref), (name, (start, stop, step), (G, W, cap, mr), policy)


@pytest.mark.parametrize("epi,omega", [(1, 1.0), (2, 0.7), (2, 1.3)])
def test_replay_other_updates(emul, epi, omega):
    A = OPS["symrand"]
    n = A.shape[0]
    rng = np.random.RandomState(2)
    x, b = rng.rand(n), rng.rand(n)
    for (start, stop, step) in [(0, n, 1), (n - 1, -1, -1)]:
        ref = ref_sweep(A, x, b, start, stop, step, epi, omega)
        for policy in (0, 1):
            got, st = run_emul(emul, A, x, b, start, stop, step, 9, 64, 60, 16, epi=epi, omega=omega, policy=policy)
            assert st[6] == 0 and st[7] == 0
            assert np.array_equal(got, ref)


def test_ring_wraps_and_far_values_go_global(emul):
    """A tiny ring forces in-tile dependencies that reach further back than the ring onto the global hand-off."""
    A = poisson_csr((40, 40))
    n = A.shape[0]
    rng = np.random.RandomState(3)
    x, b = rng.rand(n), rng.rand(n)
    ref = ref_sweep(A, x, b, 0, n, 1)
    got, st = run_emul(emul, A, x, b, 0, n, 1, 2, 64, 1020, 64, policy=1)
    assert st[4] > 0 and st[3] > 0          # both kinds of early entries occur
    assert st[6] == 0 and st[7] == 0
    assert np.array_equal(got, ref)
    # one tile, ring larger than the operator: nothing is published
    got, st = run_emul(emul, A, x, b, 0, n, 1, 1, 2048, 1020, 64)
    assert st[4] == 0 and st[5] == 0
    assert np.array_equal(got, ref)


def test_plan_statistics_of_a_stencil(emul):
    """3-D 7-point stencil cut into z-slab chunks: 2/3 of the early entries stay inside a tile."""
    A = poisson_csr((16, 16, 16))
    n = A.shape[0]
    x, b = np.ones(n), np.ones(n)
    _, st = run_emul(emul, A, x, b, 0, n, 1, 16, 2048, 1020, 64)
    assert st[0] == 16 and st[2] == 46       # 16 tiles, 3*16-2 dependency levels
    # chunks are balanced by work, so they are z-planes up to a few rows: about one z-neighbour per row crosses a
    # tile boundary, the x- and y-neighbours stay local
    assert 15 * 256 <= st[4] <= 15 * 256 + 16 * 32
    assert st[3] + st[4] == 3 * 16 * 16 * 15


def test_pencil_tiles_on_a_grid_stencil(emul):
    """Partition 1: a three-band stencil on a lexicographic grid is cut into pencils (all of x, ty lines, tz planes);
    far fewer early entries cross a tile boundary than with contiguous chunks, and every replay is still bit-exact."""
    A = poisson_csr((12, 16, 10))
    n = A.shape[0]
    rng = np.random.RandomState(4)
    x, b = rng.rand(n), rng.rand(n)
    for (start, stop, step) in [(0, n, 1), (n - 1, -1, -1)]:
        ref = ref_sweep(A, x, b, start, stop, step)
        cross = {}
        for part in (0, 1):
            for policy in (0, 1, 2):
                got, st = run_emul(emul, A, x, b, start, stop, step, 12, 256, 508, 64, policy=policy, partition=part)
                assert st[6] == 0 and st[7] == 0, (part, policy, st)
                assert np.array_equal(got, ref), (part, policy, (start, stop, step))
            cross[part] = st[4]
        assert cross[1] < cross[0]
    # anything that is not such a stencil falls back to contiguous chunks
    got, st = run_emul(emul, OPS["symrand"], np.ones(400), np.ones(400), 0, 400, 1, 7, 256, 508, 64, partition=1)
    assert st[0] == 7 and st[7] == 0
