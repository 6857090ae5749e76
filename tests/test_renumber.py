"""Renumbering of interior levels (pyamg_amd/renumber.py, csrc/pamg_renumber.hip): host logic, no GPU."""
import dataclasses

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN
from pyamg_amd import _capi as capi
from pyamg_amd import renumber as RN
from pyamg_amd.hierarchy import SparseOp, load_spec


def _random_csr(rng, m, n, dtype, density=0.2):
    M = sp.random(m, n, density=density, random_state=rng, format="csr", dtype=np.float64).astype(dtype)
    # unsorted columns inside the rows and a few empty rows: the stored order is what has to survive
    ip, ix, dx = M.indptr.astype(np.int32), M.indices.astype(np.int32).copy(), M.data.copy()
    for i in range(m):
        p = rng.permutation(ip[i + 1] - ip[i])
        ix[ip[i]:ip[i + 1]] = ix[ip[i]:ip[i + 1]][p]
        dx[ip[i]:ip[i + 1]] = dx[ip[i]:ip[i + 1]][p]
    return SparseOp("csr", (m, n), (1, 1), ip, ix, dx)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csr_renumber_moves_rows_renames_columns_keeps_stored_order(dtype):
    rng = np.random.RandomState(3)
    op = _random_csr(rng, 57, 43, dtype)
    rows, cols = rng.permutation(57).astype(np.int32), rng.permutation(43).astype(np.int32)
    for rp, cp in ((rows, cols), (rows, None), (None, cols)):
        got = RN.renumber_op(op, rp, cp)
        src = np.arange(57) if rp is None else rp
        for i in range(57):
            a, b = op.indptr[src[i]], op.indptr[src[i] + 1]
            c, d = got.indptr[i], got.indptr[i + 1]
            want_cols = op.indices[a:b] if cp is None else cp[op.indices[a:b]]
            assert d - c == b - a and np.array_equal(got.indices[c:d], want_cols) and np.array_equal(got.data[c:d], op.data[a:b])
    assert RN.renumber_op(op, None, None) is op


def test_csr_renumber_refuses_what_is_not_a_permutation():
    rng = np.random.RandomState(4)
    op = _random_csr(rng, 20, 20, np.float64)
    lib = capi.load()
    ip, ix, dx = op.indptr, op.indices, op.data
    Bp, Bj, Bx = np.empty(21, np.int32), np.empty(ix.size, np.int32), np.empty(dx.size)
    lens = np.diff(ip)
    i, j = int(np.argmax(lens)), int(np.argmin(lens))
    assert lens[i] != lens[j]
    bad = np.arange(20, dtype=np.int32)
    bad[j] = i                                               # row i twice, row j never: the lengths no longer add up
    st = lib.pamg_csr_renumber(capi.F64, 20, 20, capi.ptr(ip), capi.ptr(ix), capi.ptr(dx), capi.ptr(bad), None, capi.ptr(Bp), capi.ptr(Bj), capi.ptr(Bx))
    assert st == capi.E_ARG
    bad[j] = 20
    st = lib.pamg_csr_renumber(capi.F64, 20, 20, capi.ptr(ip), capi.ptr(ix), capi.ptr(dx), capi.ptr(bad), None, capi.ptr(Bp), capi.ptr(Bj), capi.ptr(Bx))
    assert st == capi.E_ARG
    ix2 = ix.copy(); ix2[0] = 20                             # a column outside the operator
    ident = np.arange(20, dtype=np.int32)
    st = lib.pamg_csr_renumber(capi.F64, 20, 20, capi.ptr(ip), capi.ptr(ix2), capi.ptr(dx), None, capi.ptr(ident), capi.ptr(Bp), capi.ptr(Bj), capi.ptr(Bx))
    assert st == capi.E_ARG


def test_row_argmax_abs():
    rng = np.random.RandomState(5)
    op = _random_csr(rng, 64, 31, np.float64, density=0.15)
    op.data[:] = rng.randn(op.data.size)
    got = RN._group_of(op)
    for i in range(64):
        a, b = op.indptr[i], op.indptr[i + 1]
        want = 0 if a == b else op.indices[a + int(np.argmax(np.abs(op.data[a:b])))]
        assert got[i] == want


@pytest.mark.parametrize("name", ["sa2d_cheby", "sa2d_jacobi", "rs2d_jacobi", "rs3d_jacobi_f32", "sa2d_richardson_W"])
def test_renumbered_hierarchy_is_the_same_hierarchy_permuted(name):
    spec = load_spec(GOLDEN / f"hier_{name}.npz")[0]
    dev, orders = RN.renumber_levels(spec, min_rows=8)
    nlev = len(spec.levels)
    assert sorted(orders) == [l for l in range(1, nlev - 1) if spec.levels[l].A.shape[0] >= 8] and 0 not in orders and nlev - 1 not in orders
    rng = np.random.RandomState(0)
    for l in range(nlev):
        L, D = spec.levels[l], dev.levels[l]
        o_row = orders.get(l)
        n = L.A.shape[0]
        o_row = np.arange(n) if o_row is None else o_row
        assert np.array_equal(np.sort(o_row), np.arange(n))
        x = rng.rand(n).astype(L.A.dtype)
        # row sums in stored order on both sides (scipy's csr_matvec walks a row front to back): bit-identical, permuted
        assert np.array_equal(sp.csr_array(D.A.to_scipy()) @ x[o_row], (sp.csr_array(L.A.to_scipy()) @ x)[o_row])
        assert D.A.fmt == L.A.fmt and D.pre is L.pre and D.post is L.post
        if l < nlev - 1:
            o_col = orders.get(l + 1)
            nc = L.P.shape[1]
            o_col = np.arange(nc) if o_col is None else o_col
            xc = rng.rand(nc).astype(L.A.dtype)
            assert np.array_equal(sp.csr_array(D.P.to_scipy()) @ xc[o_col], (sp.csr_array(L.P.to_scipy()) @ xc)[o_row])
            assert np.array_equal(sp.csr_array(D.R.to_scipy()) @ x[o_row], (sp.csr_array(L.R.to_scipy()) @ x)[o_col])
    # the caller's spec is untouched
    again = load_spec(GOLDEN / f"hier_{name}.npz")[0]
    for L, M in zip(spec.levels, again.levels):
        assert np.array_equal(L.A.indices, M.A.indices) and np.array_equal(L.A.data, M.A.data)


@pytest.mark.parametrize("name", ["sa2d_gs", "sa2d_sor", "sa2d_cg", "sa2d_schwarz", "rs2d_nonsym_gsnr", "el2d_jacobi", "air2d_fcjacobi"])
def test_levels_with_order_dependent_smoothers_or_blocks_keep_their_numbering(name):
    spec = load_spec(GOLDEN / f"hier_{name}.npz")[0]
    dev, orders = RN.renumber_levels(spec, min_rows=1)
    assert orders == {} and dev is spec


def test_nested_order_groups_unknowns_by_their_next_level_aggregate():
    spec = load_spec(GOLDEN / "hier_sa2d_jacobi.npz")[0]
    orders = RN.nested_orders(spec, [1])
    g = RN._group_of(spec.levels[1].P)[orders[1]]
    # every aggregate of the next level is one contiguous run of the new numbering
    assert np.count_nonzero(np.diff(g)) == np.unique(g).size - 1
    # a mixed hierarchy: only the eligible level is renumbered, and its neighbours' transfer operators follow
    mixed = dataclasses.replace(spec, levels=[spec.levels[0], spec.levels[1], dataclasses.replace(spec.levels[2], pre=dataclasses.replace(spec.levels[2].pre, kind="gauss_seidel")), spec.levels[3]])
    dev, orders = RN.renumber_levels(mixed, min_rows=8)
    assert sorted(orders) == [1]
    assert dev.levels[2].A is mixed.levels[2].A and dev.levels[1].A is not mixed.levels[1].A


@pytest.mark.parametrize("fmt,dtype", [("csr", np.float64), ("csr", np.float32), ("bsr", np.float64)])
def test_sort_rows_is_scipys_sort_indices(fmt, dtype):
    from pyamg_amd.aggregation import _sort_indices
    rng = np.random.RandomState(7)
    op = _random_csr(rng, 300, 200, dtype, density=0.1)
    if fmt == "csr":
        a = sp.csr_matrix((op.data.copy(), op.indices.copy(), op.indptr.copy()), shape=op.shape)
        b = sp.csr_matrix((op.data.copy(), op.indices.copy(), op.indptr.copy()), shape=op.shape)
    else:
        blocks = rng.rand(op.indices.size, 2, 3).astype(dtype)
        a = sp.bsr_matrix((blocks.copy(), op.indices.copy(), op.indptr.copy()), shape=(600, 600))
        b = sp.bsr_matrix((blocks.copy(), op.indices.copy(), op.indptr.copy()), shape=(600, 600))
    assert not a.has_sorted_indices
    _sort_indices(a)
    b.sort_indices()
    assert a.has_sorted_indices and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data) and np.array_equal(a.indptr, b.indptr)
    _sort_indices(a)                                   # sorted already: nothing moves
    assert np.array_equal(a.indices, b.indices)


def test_host_cpus_honours_the_cgroup_quota(tmp_path, monkeypatch):
    """csrc/pamg_host_threads.h: planners size their thread pools by what the container OWNS (the pool's GPU boxes: 256 hardware threads, a quota of 16 cores)"""
    import os
    lib = capi.load()
    f = tmp_path / "cpu.max"
    monkeypatch.delenv("PAMG_HOST_THREADS", raising=False)
    machine = min(os.cpu_count(), len(os.sched_getaffinity(0)))
    for text, want in (("1600000 100000\n", min(16, machine)), ("max 100000\n", machine), ("150000 100000\n", min(2, machine)), ("garbage", machine), ("50000 100000", 1)):
        f.write_text(text)
        monkeypatch.setenv("PAMG_CGROUP_CPU_MAX", str(f))
        assert lib.pamg_host_cpus(1) == want, text
    monkeypatch.setenv("PAMG_HOST_THREADS", "5")
    assert lib.pamg_host_cpus(1) == 5
    assert lib.pamg_host_cpus(0) >= 1
