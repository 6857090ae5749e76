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


def cube_tet_mesh(N):
    """Structured tetrahedral mesh of the unit cube: N^3 vertices, every hexahedral cell split
    into 6 tets (Kuhn triangulation).  Input for the reference's gallery.linear_elasticity_p1
    (BASELINE config 5: the reference ships no 3-D mesher, pyamg/gallery/elasticity.py:53-56)."""
    g = np.linspace(0.0, 1.0, N)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    V = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    idx = np.arange(N ** 3).reshape(N, N, N)
    c = idx[:-1, :-1, :-1].ravel()
    dx, dy, dz = N * N, N, 1
    corners = {(0, 0, 0): c, (1, 0, 0): c + dx, (0, 1, 0): c + dy, (0, 0, 1): c + dz, (1, 1, 0): c + dx + dy,
               (1, 0, 1): c + dx + dz, (0, 1, 1): c + dy + dz, (1, 1, 1): c + dx + dy + dz}
    import itertools
    tets = []
    for perm in itertools.permutations(range(3)):
        p = [(0, 0, 0)]
        cur = [0, 0, 0]
        for ax in perm:
            cur[ax] = 1
            p.append(tuple(cur))
        tets.append(np.stack([corners[q] for q in p], axis=1))
    T = np.concatenate(tets, axis=0)
    # consistent (positive) orientation
    d = V[T[:, 1:]] - V[T[:, :1]]
    neg = np.linalg.det(d) < 0
    T[neg, 2], T[neg, 3] = T[neg, 3].copy(), T[neg, 2].copy()
    return V, T


def elasticity3d(N):
    """3-D linear elasticity (P1 tets) on an N^3-vertex cube, clamped on the face x = 0:
    returns (A as BSR(3,3) with int32 indices, B rigid-body modes).  Needs the reference
    (oracle/_ref) for the element assembly."""
    import pyamg
    V, E = cube_tet_mesh(N)
    A, B = pyamg.gallery.linear_elasticity_p1(V, E, format="csr")
    free = np.repeat(V[:, 0] > 0.0, 3)
    keep = np.flatnonzero(free)
    A = sp.csr_array(A[keep][:, keep])
    B = B[keep]
    A.indptr = A.indptr.astype(np.int32)
    A.indices = A.indices.astype(np.int32)
    Ab = A.tobsr(blocksize=(3, 3))
    Ab.indptr = Ab.indptr.astype(np.int32)
    Ab.indices = Ab.indices.astype(np.int32)
    return Ab, B
