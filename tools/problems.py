"""Synthetic bench/test inputs (NOT product code): vectorised generator of the
reference's model problem ``pyamg.gallery.poisson(grid, format='csr')`` (reference:
pyamg/gallery/laplacian.py:10-79 via stencil_grid) -- same matrix, built without the
reference so that kernel micro-benchmarks do not need it."""
import numpy as np
import scipy.sparse as sp


def poisson_csr(grid, dtype=np.float64):
    """N-D finite-difference Laplacian, lexicographic ordering (last axis fastest), CSR
    with sorted column indices, int32 index arrays."""
    grid = tuple(int(g) for g in grid)
    nd = len(grid)
    n = int(np.prod(grid))
    strides = [int(np.prod(grid[k + 1:])) for k in range(nd)]
    idx = np.arange(n, dtype=np.int64)
    coords = [(idx // strides[k]) % grid[k] for k in range(nd)]
    # candidate neighbours in ascending column order: -s0, -s1, ..., 0, ..., +s1, +s0
    offs, valid = [], []
    for k in range(nd):
        offs.append(-strides[k]); valid.append(coords[k] > 0)
    offs.append(0); valid.append(np.ones(n, dtype=bool))
    for k in reversed(range(nd)):
        offs.append(strides[k]); valid.append(coords[k] < grid[k] - 1)
    V = np.stack(valid, axis=1)                              # (n, 2nd+1)
    counts = V.sum(axis=1)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    cols = (idx[:, None] + np.array(offs, dtype=np.int64)[None, :])[V]
    vals = np.broadcast_to(np.array([-1.0] * nd + [2.0 * nd] + [-1.0] * nd, dtype=dtype)[None, :], V.shape)[V]
    A = sp.csr_array((np.ascontiguousarray(vals), cols.astype(np.int32), indptr.astype(np.int32)), shape=(n, n))
    return A


def spmv_bytes(A):
    """Algorithmic bytes of y = A x: values + column ids + row pointer +
    x read once + y written once."""
    n_rows, n_cols = A.shape
    vb = A.dtype.itemsize
    return (vb + 4) * A.nnz + 4 * (n_rows + 1) + vb * n_cols + vb * n_rows
