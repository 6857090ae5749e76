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


def elasticity_p1_batched(V, T, E=1e5, nu=0.3, keep=None):
    """The operator and rigid-body modes of the reference's ``gallery.linear_elasticity_p1`` for tets (3-D P1 elements,
    pyamg/gallery/elasticity.py:209-351: K_e = det/6 * R^T C R with the basis-function gradients in R), with the element
    loop -- a Python loop in the reference, minutes for a million elements -- batched through NumPy.  Workload
    generator only: the engines under test get the same matrix either way (checked against the reference's own
    assembly in tests/test_host.py).  ``keep``: vertex ids that stay (the others are clamped and dropped).  Returns the
    operator as BSR(3,3) and the six rigid-body modes of the kept vertices."""
    lame = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 + 2 * nu)
    V = np.asarray(V, dtype=np.float64)
    T = np.asarray(T)
    ne = T.shape[0]
    M = np.ones((ne, 4, 4))
    M[:, 1:, :] = np.swapaxes(V[T], 1, 2)                    # rows: 1, x, y, z of the four vertices
    grad = np.linalg.inv(M)[:, :, 1:]                        # (ne, 4 vertices, 3): gradients of the basis functions
    det = np.linalg.det(M)
    # K_e[(a,i),(b,j)] = det/6 * (lame g_a[i] g_b[j] + mu g_a[j] g_b[i] + mu delta_ij g_a.g_b): the closed form of the
    # reference's R^T C R (C = lame 1 1^T + 2 mu I on the normal strains, mu on the engineering shear strains)
    # everything below keeps the ELEMENT index as the contiguous (last) axis: long vector operations only
    G = np.ascontiguousarray(np.transpose(grad, (2, 1, 0)))             # (3 components, 4 vertices, ne)
    w = det / 6.0
    dots = np.empty((4, 4, ne))
    for a_ in range(4):
        for b_ in range(4):
            dots[a_, b_] = mu * w * (G[0, a_] * G[0, b_] + G[1, a_] * G[1, b_] + G[2, a_] * G[2, b_])
    lw, mw = lame * w, mu * w
    nv = V.shape[0]
    vi = np.repeat(T.T[:, None, :], 4, axis=1).reshape(-1).astype(np.int64)     # (a, b, e) order: vertex a of element e
    vj = np.repeat(T.T[None, :, :], 4, axis=0).reshape(-1).astype(np.int64)
    sel = None
    if keep is not None:                                     # drop clamped vertices before the sort
        new_id = np.full(nv, -1, dtype=np.int64)
        new_id[keep] = np.arange(keep.size)
        vi, vj = new_id[vi], new_id[vj]
        sel = np.flatnonzero((vi >= 0) & (vj >= 0))
        vi, vj = vi[sel], vj[sel]
        nv = keep.size
    key = vi * nv + vj
    order = np.argsort(key, kind="stable")
    key = key[order]
    first = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    take = order if sel is None else sel[order]
    data = np.empty((first.size, 3, 3))
    plane = np.empty((4, 4, ne))
    for i in range(3):
        for j in range(3):
            # K_e[(a,i),(b,j)] = det/6 * (lame g_a[i] g_b[j] + mu g_a[j] g_b[i] + mu delta_ij g_a.g_b): the closed form of
            # the reference's R^T C R (C = lame 1 1^T + 2 mu I on normal strains, mu on engineering shear strains)
            for a_ in range(4):
                for b_ in range(4):
                    np.multiply(G[i, a_], G[j, b_], out=plane[a_, b_])
                    plane[a_, b_] *= lw
                    plane[a_, b_] += mw * G[j, a_] * G[i, b_]
            if i == j:
                plane += dots
            data[:, i, j] = np.add.reduceat(plane.reshape(-1)[take], first)    # duplicates of a vertex pair summed
    ukey = key[first]
    rows_b, cols_b = ukey // nv, ukey % nv
    indptr = np.zeros(nv + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(np.bincount(rows_b, minlength=nv))
    A = sp.bsr_array((data, cols_b.astype(np.int32), indptr.astype(np.int32)), shape=(3 * nv, 3 * nv))
    Vk = V if keep is None else V[keep]
    n = 3 * nv
    B = np.zeros((n, 6))
    B[0::3, 0] = 1; B[1::3, 1] = 1; B[2::3, 2] = 1
    B[0::3, 3] = -Vk[:, 1]; B[1::3, 3] = Vk[:, 0]
    B[0::3, 4] = -Vk[:, 2]; B[2::3, 4] = Vk[:, 0]
    B[1::3, 5] = -Vk[:, 2]; B[2::3, 5] = Vk[:, 1]
    return A, B


def elasticity3d(N):
    """3-D linear elasticity (P1 tets) on an N^3-vertex cube, clamped on the face x = 0:
    returns (A as BSR(3,3) with int32 indices, B rigid-body modes)."""
    V, E = cube_tet_mesh(N)
    Ab, B = elasticity_p1_batched(V, E, keep=np.flatnonzero(V[:, 0] > 0.0))
    return Ab, B
