"""GPU parity, setup-phase operators: pyamg_amd.aggregation against SciPy (whose sparse products ARE
what the reference's setup computes: aggregation.py:425, smooth.py:199) and against the live reference (oracle/_ref).

Bars: sparse products and differences are THE ARRAYS SciPy produces -- same stored order (its linked-list emission
order), same pattern (exact zeros dropped like SciPy drops them), same bits; spectral radius within 1e-10
relative of the reference's for the same random stream (only the dot products' summation order differs); smoothed
prolongators and coarse operators within 1e-13 relative (they inherit the spectral radius' last bits)."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _shuffle_rows(M, rng):
    """same matrix, stored entries of every row in random order (SciPy's products depend on that order)"""
    M = M.tocsr().copy()
    for i in range(M.shape[0]):
        lo, hi = M.indptr[i], M.indptr[i + 1]
        p = rng.permutation(hi - lo)
        M.indices[lo:hi] = M.indices[lo:hi][p]
        M.data[lo:hi] = M.data[lo:hi][p]
    M.has_sorted_indices = False
    return M


def _sorted(M):
    M = M.tocsr().copy()
    M.sort_indices()
    return M


def _same(C, ref):
    """the very arrays SciPy produced: same stored order, same pattern, same bits"""
    assert C.shape == ref.shape and C.nnz == ref.nnz and C.format == ref.format
    assert np.array_equal(C.indptr, ref.indptr) and np.array_equal(C.indices, ref.indices)
    assert np.array_equal(np.ravel(C.data), np.ravel(ref.data))


def test_matmat_bit_exact_against_scipy():
    from pyamg_amd.aggregation import DeviceCSR
    rng = np.random.default_rng(5)
    for (m, k, n, da, db) in ((300, 200, 250, 0.05, 0.05), (1, 1, 1, 1.0, 1.0), (50, 70, 3, 0.3, 0.5), (2000, 2000, 2000, 0.004, 0.004)):
        A = _shuffle_rows(sp.random_array((m, k), density=da, random_state=rng, format="csr"), rng)
        B = sp.random_array((k, n), density=db, random_state=rng, format="csr")
        Ad, Bd = DeviceCSR.from_scipy(A), DeviceCSR.from_scipy(B)
        C = (Ad @ Bd).to_scipy()
        _same(C, A @ B)
    # empty rows / empty operands
    Z = sp.csr_array((40, 40))
    _same((DeviceCSR.from_scipy(Z) @ DeviceCSR.from_scipy(Z)).to_scipy(), Z @ Z)
    E = sp.csr_array(np.eye(40))
    E = sp.csr_array((E.data[:20], E.indices[:20], np.r_[E.indptr[:21], [20] * 20]), shape=(40, 40))
    _same((DeviceCSR.from_scipy(E) @ DeviceCSR.from_scipy(E)).to_scipy(), E @ E)


def test_matmat_drops_exact_zeros_like_scipy():
    from pyamg_amd.aggregation import DeviceCSR
    A = sp.csr_array(np.array([[1.0, -1.0, 0.0], [2.0, 0.0, 1.0], [0.0, 0.0, 0.0]]))
    B = sp.csr_array(np.array([[3.0, 1.0], [3.0, 0.0], [-6.0, 5.0]]))
    ref = A @ B
    C = (DeviceCSR.from_scipy(A) @ DeviceCSR.from_scipy(B)).to_scipy()
    assert ref.nnz == 2
    _same(C, ref)


def test_matmat_long_rows():
    """rows with more than 8192 products (every row of a coarse Galerkin product) take the dense-accumulator path:
    windows of 2048 output columns, batches in sequence order, SciPy's emission order restored per finished row"""
    from pyamg_amd.aggregation import DeviceCSR
    rng = np.random.default_rng(6)
    m, k, n = 64, 3000, 5000
    A = sp.random_array((m, k), density=0.002, random_state=rng, format="lil")
    A[3, :] = rng.standard_normal(k)                        # 3000 entries x ~10 each = 30 000 products
    A[40, ::2] = rng.standard_normal(k // 2)
    A = _shuffle_rows(A.tocsr(), rng)
    B = sp.random_array((k, n), density=0.002, random_state=rng, format="csr")
    B.sort_indices()
    C = (DeviceCSR.from_scipy(A) @ DeviceCSR.from_scipy(B)).to_scipy()
    _same(C, A @ B)
    Bu = _shuffle_rows(B, rng)                              # unsorted right operand
    _same((DeviceCSR.from_scipy(A) @ DeviceCSR.from_scipy(Bu)).to_scipy(), A @ Bu)
    # every row long, few distinct output columns (the shape of (R A) P on a coarse level), exact cancellations
    m, k, n = 40, 900, 300
    A2 = _shuffle_rows(sp.random_array((m, k), density=0.6, random_state=rng, format="csr"), rng)
    B2 = sp.random_array((k, n), density=0.09, random_state=rng, format="csr")
    B2.data = np.sign(B2.data - 0.5)                        # +-1: sums cancel exactly now and then
    A2.data = np.round(A2.data * 4.0)
    A2.eliminate_zeros()
    ref = A2 @ B2
    assert np.diff(ref.indptr).max() <= n and (A2 @ abs(B2)).nnz > ref.nnz
    _same((DeviceCSR.from_scipy(A2) @ DeviceCSR.from_scipy(B2)).to_scipy(), ref)
    # rows of 4096 .. 8192 products: whole-row tasks of the big variant (136 KB of LDS), no windows
    B4 = sp.random_array((k, n), density=0.04, random_state=rng, format="csr")
    B4.data = np.sign(B4.data - 0.5)
    ref4 = A2 @ B4
    _same((DeviceCSR.from_scipy(A2) @ DeviceCSR.from_scipy(B4)).to_scipy(), ref4)
    B4u = _shuffle_rows(B4, rng)
    _same((DeviceCSR.from_scipy(A2) @ DeviceCSR.from_scipy(B4u)).to_scipy(), A2 @ B4u)


def test_subtract_bit_exact_against_scipy():
    from pyamg_amd.aggregation import DeviceCSR
    rng = np.random.default_rng(7)
    A = sp.random_array((400, 300), density=0.05, random_state=rng, format="csr")
    B = sp.random_array((400, 300), density=0.05, random_state=rng, format="csr")
    B = (B + 0.5 * A).tocsr()                               # shared pattern + own entries
    B.data[::7] = 0.0                                       # explicit zeros in an operand
    A.sort_indices(); B.sort_indices()
    C = (DeviceCSR.from_scipy(A) - DeviceCSR.from_scipy(B)).to_scipy()
    _same(C, A - B)
    D = (DeviceCSR.from_scipy(A) - DeviceCSR.from_scipy(A.copy())).to_scipy()     # everything cancels: empty result
    assert D.nnz == 0 and D.shape == A.shape
    # an operand with unsorted rows: SciPy switches to its general algorithm (another emission order)
    Bu = _shuffle_rows(B, rng)
    _same((DeviceCSR.from_scipy(A) - DeviceCSR.from_scipy(Bu)).to_scipy(), A - Bu)
    Au = _shuffle_rows(A, rng)
    _same((DeviceCSR.from_scipy(Au) - DeviceCSR.from_scipy(B)).to_scipy(), Au - B)


def test_galerkin_product_matches_scipy_expression():
    from pyamg_amd.aggregation import galerkin_product
    rng = np.random.default_rng(8)
    n, nc = 900, 120
    A = sp.random_array((n, n), density=0.01, random_state=rng, format="csr")
    A = (A + A.T + 4.0 * sp.eye_array(n)).tocsr()
    P = sp.random_array((n, nc), density=0.03, random_state=rng, format="csr")
    R = P.T.tocsr()
    _same(galerkin_product(R, A, P), R @ A @ P)
    # the formats of an SA hierarchy: BSR(1,1) restriction / prolongation, CSR operator -> BSR(1,1) coarse operator
    Pb, Rb = sp.bsr_array(P, blocksize=(1, 1)), sp.bsr_array(R, blocksize=(1, 1))
    Ac = galerkin_product(Rb, A, Pb)
    ref = Rb @ A @ Pb
    assert Ac.format == ref.format == "bsr" and tuple(Ac.blocksize) == tuple(ref.blocksize) == (1, 1)
    _same(Ac, ref)
    # true blocks (elasticity-like): 2x2 fine blocks, 2x3 prolongation blocks
    nb, ncb = 150, 20
    Ab = sp.random_array((nb, nb), density=0.05, random_state=rng, format="csr")
    Ab = sp.kron(Ab + Ab.T + 4.0 * sp.eye_array(nb), np.array([[2.0, 0.5], [0.5, 3.0]]), format="bsr")
    Ab = sp.bsr_array(Ab, blocksize=(2, 2))
    Pk = sp.bsr_array(sp.kron(sp.random_array((nb, ncb), density=0.1, random_state=rng, format="csr"),
                              rng.standard_normal((2, 3)), format="bsr"), blocksize=(2, 3))
    Rk = sp.bsr_array(Pk.T.tocsr(), blocksize=(3, 2))
    Acb = galerkin_product(Rk, Ab, Pk)
    refb = Rk @ Ab @ Pk
    assert Acb.format == "bsr" and tuple(Acb.blocksize) == (3, 3)
    _same(Acb, refb)                                        # block order and the zeros inside blocks included
    with pytest.raises(NotImplementedError):
        galerkin_product(R, sp.bsr_array(A, blocksize=(2, 2)), P)


def test_products_with_a_unit_outer_block_and_a_true_inner_block():
    """SA on a BSR operator with ONE candidate: R has (1,3) blocks, P (3,1).  SciPy's bsr_matmat takes the csr_matmat
    shortcut (reverse first-touch order, zeros dropped) only for R == N == C == 1; (1,3) @ (3,1) and (1,3) @ (3,3) are
    stored block-wise -- forward first-touch order, zeros kept -- although the result has 1x1 blocks."""
    from pyamg_amd.aggregation import DeviceCSR, _device_product, galerkin_product
    rng = np.random.default_rng(23)
    nb, ncb = 90, 25
    Ab = sp.random_array((nb, nb), density=0.06, random_state=rng, format="csr")
    Ab = sp.bsr_array(sp.kron(Ab + Ab.T + 4.0 * sp.eye_array(nb), np.array([[2.0, 0.5, 0.0], [0.5, 3.0, 1.0], [0.0, 1.0, 4.0]]), format="bsr"),
                      blocksize=(3, 3))
    Pk = sp.bsr_array(sp.kron(sp.random_array((nb, ncb), density=0.12, random_state=rng, format="csr"),
                              np.array([[1.0], [-2.0], [0.5]]), format="bsr"), blocksize=(3, 1))
    Rk = sp.bsr_array(Pk.T.tocsr(), blocksize=(1, 3))
    ref1 = Rk @ Pk                                           # (1,3) @ (3,1) -> 1x1 blocks, stored block-wise
    assert ref1.format == "bsr" and tuple(ref1.blocksize) == (1, 1)
    _same(_device_product(Rk, Pk), ref1)
    ref = Rk @ Ab @ Pk
    Ac = galerkin_product(Rk, Ab, Pk)
    assert Ac.format == "bsr" and tuple(Ac.blocksize) == (1, 1)
    _same(Ac, ref)
    # a product whose exact zeros the block-wise path must KEEP: rows of R orthogonal to the matching block of P
    Pz = sp.bsr_array((np.array([[[1.0], [1.0], [0.0]], [[2.0], [0.0], [1.0]]]), np.array([0, 1]), np.array([0, 1, 2])), shape=(6, 2))
    Rz = sp.bsr_array((np.array([[[1.0, -1.0, 5.0]], [[0.0, 3.0, 0.0]]]), np.array([0, 1]), np.array([0, 1, 2])), shape=(2, 6))
    refz = Rz @ Pz
    assert refz.nnz == 2 and refz.data.ravel()[0] == 0.0 and refz.data.ravel()[1] == 0.0
    _same(_device_product(Rz, Pz), refz)


def _reference():
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present on this box")
    import pyamg
    return pyamg


def test_spectral_radius_against_the_reference():
    pyamg = _reference()
    from pyamg.util.linalg import approximate_spectral_radius as ref_rho
    from pyamg_amd.aggregation import approximate_spectral_radius
    cases = []
    A = pyamg.gallery.poisson((60, 60), format="csr")
    cases.append(A)
    cases.append(pyamg.gallery.poisson((20, 20, 20), format="csr"))
    # D^-1 A scaled copy as the prolongation smoother forms it, BSR(1,1) storage
    D = A.diagonal()
    cases.append(sp.bsr_array(sp.csr_array(A.multiply(1.0 / D[:, None])), blocksize=(1, 1)))
    # non-symmetric with a complex dominant pair: the restart vector becomes complex (planes = 2 on the device)
    rng = np.random.default_rng(3)
    n = 400
    K = sp.diags_array([np.ones(n - 1), -np.ones(n - 1)], offsets=[1, -1], format="csr") * 3.0
    cases.append((K + sp.random_array((n, n), density=0.01, random_state=rng, format="csr") * 0.1 + 0.05 * sp.eye_array(n)).tocsr())
    for k, M in enumerate(cases):
        for kw in ({}, {"maxiter": 8, "restart": 2}, {"tol": 1e-6, "maxiter": 20, "restart": 8}):
            M1, M2 = M.copy(), M.copy()
            np.random.seed(17 + k)
            r_ref = ref_rho(M1, **kw)
            np.random.seed(17 + k)
            r_dev = approximate_spectral_radius(M2, **kw)
            assert abs(r_dev - r_ref) <= 1e-10 * abs(r_ref), (k, kw, r_dev, r_ref)
            assert M2.rho == r_dev and approximate_spectral_radius(M2) == r_dev        # cached on the matrix
    # the random stream is consumed exactly like the reference consumes it
    np.random.seed(5); ref_rho(cases[0].copy()); a = np.random.rand()
    np.random.seed(5); approximate_spectral_radius(cases[0].copy()); b = np.random.rand()
    assert a == b
    # return_vector / initial_guess
    v0 = np.random.rand(cases[0].shape[0], 1)
    r1, v1 = ref_rho(cases[0].copy(), initial_guess=v0.copy(), return_vector=True)
    r2, v2 = approximate_spectral_radius(cases[0].copy(), initial_guess=v0.copy(), return_vector=True)
    assert abs(r1 - r2) <= 1e-10 * r1 and v2.shape == v1.shape
    assert np.linalg.norm(v2 - v1) <= 1e-8 * np.linalg.norm(v1)
    # error behaviour
    with pytest.raises(ValueError):
        approximate_spectral_radius(sp.csr_array((3, 4)))
    with pytest.raises(ValueError):
        approximate_spectral_radius(cases[0].copy(), maxiter=0)


def test_prolongation_smoothers_against_the_reference():
    pyamg = _reference()
    from pyamg.aggregation.smooth import jacobi_prolongation_smoother as ref_jac, richardson_prolongation_smoother as ref_rich
    from pyamg.aggregation.aggregate import standard_aggregation
    from pyamg.aggregation.tentative import fit_candidates
    from pyamg.strength import symmetric_strength_of_connection
    from pyamg_amd.aggregation import jacobi_prolongation_smoother, richardson_prolongation_smoother
    for grid in ((50, 50), (16, 16, 16)):
        A = pyamg.gallery.poisson(grid, format="csr")
        Cs = symmetric_strength_of_connection(A)
        AggOp, _ = standard_aggregation(Cs)
        T, _ = fit_candidates(AggOp, np.ones((A.shape[0], 1)))
        for kw in ({}, {"degree": 2}, {"weighting": "local"}, {"omega": 1.0, "weighting": "block"}):
            np.random.seed(3)
            Pr = ref_jac(A.copy(), T, Cs, np.ones((A.shape[0], 1)), **kw)
            np.random.seed(3)
            Pd = jacobi_prolongation_smoother(A.copy(), T, Cs, np.ones((A.shape[0], 1)), **kw)
            assert Pd.format == Pr.format and tuple(Pd.blocksize) == tuple(Pr.blocksize)
            assert np.array_equal(Pd.indptr, Pr.indptr) and np.array_equal(Pd.indices, Pr.indices)      # stored order too
            assert np.max(np.abs(np.ravel(Pd.data) - np.ravel(Pr.data))) <= 1e-13 * np.max(np.abs(Pr.data))
        np.random.seed(4)
        Pr = ref_rich(A.copy(), T)
        np.random.seed(4)
        Pd = richardson_prolongation_smoother(A.copy(), T)
        assert np.array_equal(Pd.indices, Pr.indices)
        assert np.max(np.abs(np.ravel(Pd.data) - np.ravel(Pr.data))) <= 1e-13 * np.max(np.abs(Pr.data))
    with pytest.raises(NotImplementedError):
        jacobi_prolongation_smoother(A, T, Cs, np.ones((A.shape[0], 1)), filter_entries=True)


def test_block_prolongation_smoothing_against_the_reference():
    """prolongation smoothing of TRUE block operators (elasticity: BSR(2,2) / (3,3) operators, tentative prolongators with
    (2,3) / (3,6) blocks): weightings 'diagonal', 'local' and 'block' (inverted diagonal blocks: amg_core.pinv_array on the
    device).  SciPy runs bsr_matmat and bsr_binop_bsr there -- whole blocks in first-touch order, a block of the
    difference dropped only when every entry is zero -- so pattern and stored order must be the reference's exactly,
    the values within the spectral radius' last bits."""
    pyamg = _reference()
    from pyamg.aggregation.aggregate import standard_aggregation
    from pyamg.aggregation.smooth import jacobi_prolongation_smoother as ref_jac, richardson_prolongation_smoother as ref_rich
    from pyamg.aggregation.tentative import fit_candidates
    from pyamg.strength import symmetric_strength_of_connection as ref_strength
    from pyamg_amd.aggregation import jacobi_prolongation_smoother, richardson_prolongation_smoother
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
    from tools.problems import elasticity3d
    A2, B2 = pyamg.gallery.linear_elasticity((30, 30), format="bsr")
    A3, B3 = elasticity3d(9)
    for A, B in ((A2, B2), (A3, B3)):
        assert A.format == "bsr" and A.blocksize[0] > 1
        C = ref_strength(A, theta=0.0)
        AggOp = standard_aggregation(C)[0]
        T, Bc = fit_candidates(AggOp, B)
        assert T.format == "bsr" and tuple(T.blocksize) == (A.blocksize[0], B.shape[1])
        for kw in ({}, {"weighting": "local"}, {"weighting": "block"}, {"degree": 2, "omega": 1.0}):
            np.random.seed(7)
            ref = ref_jac(A.copy(), T, C, Bc, **kw)
            np.random.seed(7)
            P = jacobi_prolongation_smoother(A.copy(), T, C, Bc, **kw)
            assert P.format == ref.format == "bsr" and tuple(P.blocksize) == tuple(ref.blocksize), kw
            assert np.array_equal(P.indptr, ref.indptr) and np.array_equal(P.indices, ref.indices), kw
            assert np.max(np.abs(P.data - ref.data)) <= 1e-13 * np.max(np.abs(ref.data)), kw
        np.random.seed(7)
        ref = ref_rich(A.copy(), T)
        np.random.seed(7)
        P = richardson_prolongation_smoother(A.copy(), T)
        assert np.array_equal(P.indptr, ref.indptr) and np.array_equal(P.indices, ref.indices)
        assert np.max(np.abs(P.data - ref.data)) <= 1e-13 * np.max(np.abs(ref.data))


def test_block_difference_is_scipys():
    """pamg_csr_subtract_bsr against SciPy's BSR - BSR: general (unsorted) and canonical operands, blocks that cancel
    exactly (dropped), blocks with single zero entries (kept whole), empty block rows"""
    import ctypes as C
    from pyamg_amd import _capi as capi
    from pyamg_amd.aggregation import DeviceCSR
    rng = np.random.default_rng(3)
    R, Cc, nb, ncb = 3, 2, 40, 25
    def rand_bsr(density, shuffle):
        pat = sp.random_array((nb, ncb), density=density, random_state=rng, format="csr")
        pat.sort_indices()
        if shuffle:
            pat = _shuffle_rows(pat, rng)
        data = rng.integers(-2, 3, size=(pat.nnz, R, Cc)).astype(np.float64)
        return sp.bsr_array((data, pat.indices.astype(np.int32), pat.indptr.astype(np.int32)), shape=(nb * R, ncb * Cc))
    for shuffle in (False, True):
        A, B = rand_bsr(0.2, shuffle), rand_bsr(0.2, shuffle)
        # make some blocks cancel exactly and some share a column with different values
        B.data[: min(len(B.data), 30)] = 0.0
        A2 = sp.bsr_array((np.concatenate([A.data, B.data]), np.concatenate([A.indices, B.indices]),
                           A.indptr + B.indptr), shape=A.shape) if False else A
        ref = A2 - B
        same = A2 - A2.copy()
        for X, Y, want in ((A2, B, ref), (A2, A2.copy(), same)):
            Xd, Yd = DeviceCSR.from_scipy(X), DeviceCSR.from_scipy(Y)
            h = C.c_void_p()
            capi.check(capi.lib().pamg_csr_subtract_bsr(Xd.handle, Yd.handle, R, Cc, C.byref(h)), "pamg_csr_subtract_bsr")
            out = DeviceCSR(h).to_scipy(blocksize=(R, Cc))
            assert out.nnz == want.nnz, (shuffle, out.nnz, want.nnz)
            assert np.array_equal(out.indptr, want.indptr) and np.array_equal(out.indices, want.indices), shuffle
            assert np.array_equal(out.data, want.data)


def test_aggregation_and_tentative_prolongator_on_the_device():
    """f4: amg_core.standard_aggregation and fit_candidates on the device against the reference's outputs (committed) and
    the oracle: aggregates, their numbering and the C-points integer for integer; tentative-prolongator blocks and coarse
    candidates bit for bit (the per-aggregate sums run in the reference's order)"""
    from conftest import GOLDEN
    from oracle import oracle as orc
    from pyamg_amd import amg_core as gcore
    from pyamg_amd.aggregation import fit_candidates, standard_aggregation
    z = np.load(GOLDEN / "kernels_setup.npz")
    for name in ("p2d", "p3d", "irr", "aniso"):
        Ap, Aj = z[f"agg.{name}.indptr"], z[f"agg.{name}.indices"]
        n = Ap.size - 1
        x, y = np.full(n, -7, dtype=np.int32), np.full(n, -7, dtype=np.int32)
        cnt = gcore.standard_aggregation(n, Ap, Aj, x, y)
        assert cnt == z[f"agg.{name}.y"].size, name
        assert np.array_equal(x, z[f"agg.{name}.x"]) and np.array_equal(y[:cnt], z[f"agg.{name}.y"]), name
        Cg = sp.csr_array((np.ones(Aj.size), Aj, Ap), shape=(n, n))
        AggOp, Cpts = standard_aggregation(Cg)
        assert np.array_equal(Cpts, z[f"agg.{name}.y"]) and AggOp.shape == (n, cnt) and AggOp.dtype == np.int32
        keep = z[f"agg.{name}.x"] >= 0
        assert np.array_equal(AggOp.indices, z[f"agg.{name}.x"][keep]) and np.array_equal(np.diff(AggOp.indptr), keep.astype(np.int32))
    # a non-symmetric pattern is refused (the reference's third pass could open aggregates there): NotImplementedError
    Cn = sp.csr_array((np.ones(3), np.array([1, 2, 0], dtype=np.int32), np.array([0, 1, 2, 3], dtype=np.int32)), shape=(3, 3))
    with pytest.raises(NotImplementedError):
        standard_aggregation(Cn)
    Tp, Tj, shape = z["fit.Tp"], z["fit.Tj"], tuple(z["fit.shape"])
    AggOp = sp.csr_array((np.ones(Tj.size, dtype=np.int32), Tj, Tp), shape=shape)
    csc = AggOp.tocsc()
    for tag, K1, K2 in (("a", 1, 1), ("b", 2, 3), ("c", 3, 6), ("d", 1, 2)):
        B = z[f"fit.{tag}.B"]
        Q, R = fit_candidates(AggOp, B)
        assert Q.format == "bsr" and tuple(Q.blocksize) == (K1, K2) and Q.shape == (K1 * shape[0], K2 * shape[1])
        assert np.array_equal(Q.indptr, Tp) and np.array_equal(Q.indices, Tj)
        assert np.array_equal(Q.data, z[f"fit.{tag}.Q"]), tag
        assert np.array_equal(R, z[f"fit.{tag}.R"]), tag
        # the amg_core-signature entry point: CSC lists in, blocks in CSC order out
        Ax = np.zeros(Tj.size * K1 * K2, dtype=B.dtype)
        Rr = np.zeros(shape[1] * K2 * K2, dtype=B.dtype)
        gcore.fit_candidates(shape[0], shape[1], K1, K2, csc.indptr.astype(np.int32), csc.indices.astype(np.int32), Ax, np.ravel(B).copy(), Rr, 1e-10)
        Axo, Ro = orc.fit_candidates(shape[1], K1, K2, csc.indptr, csc.indices, B, 1e-10)
        assert np.array_equal(Ax, np.ravel(Axo)) and np.array_equal(Rr, np.ravel(Ro))


def test_device_aggregation_inside_the_reference_setup():
    """device_setup patches standard_aggregation and fit_candidates too: on 3-D Poisson 128^3 and 3-D elasticity 40^3 the
    reference's solver built through them has the SAME aggregates (AggOp, C-points) and tentative prolongators as the
    solver built by the reference alone (integers exact; floats <= 1e-14 relative: T and B_coarse bit-identical on the
    fine level, coarser levels inherit the last bits of the device spectral radii through the smoothed prolongator)"""
    pyamg = _reference()
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
    from tools.problems import elasticity3d
    from pyamg_amd.aggregation import device_setup
    A1 = pyamg.gallery.poisson((128, 128, 128), format="csr")
    A2, B2 = elasticity3d(40)
    for A, kw in ((A1, {}), (A2, {"B": B2, "smooth": "jacobi"})):
        np.random.seed(9)
        ref = pyamg.smoothed_aggregation_solver(A.copy(), max_coarse=10, keep=True, **kw)
        np.random.seed(9)
        with device_setup(pyamg, aggregation=True):
            dev = pyamg.smoothed_aggregation_solver(A.copy(), max_coarse=10, keep=True, **kw)
        assert len(dev.levels) == len(ref.levels)
        for l, (Ld, Lr) in enumerate(zip(dev.levels[:-1], ref.levels[:-1])):
            assert np.array_equal(Ld.AggOp.indptr, Lr.AggOp.indptr) and np.array_equal(Ld.AggOp.indices, Lr.AggOp.indices), l
            assert np.array_equal(Ld.T.indptr, Lr.T.indptr) and np.array_equal(Ld.T.indices, Lr.T.indices), l
            scale = np.max(np.abs(Lr.T.data))
            assert np.max(np.abs(Ld.T.data - Lr.T.data)) <= 1e-14 * scale, (l, np.max(np.abs(Ld.T.data - Lr.T.data)) / scale)
            Bd, Br = dev.levels[l + 1].B, ref.levels[l + 1].B
            assert np.max(np.abs(Bd - Br)) <= 1e-14 * np.max(np.abs(Br)), l
            if l == 0:
                assert np.array_equal(Ld.T.data, Lr.T.data) and np.array_equal(Bd, Br)


def test_bsr_transpose_is_scipys():
    """pamg_bsr_transpose (R = P.T of the SA setup) against SciPy's bsr_transpose: the same three arrays for 1x1 and
    true blocks, unsorted rows, empty rows and columns; inside device_products() `P.T` of a large BSR operand is routed
    there, a small one stays with SciPy"""
    import ctypes  # noqa: F401
    from pyamg_amd import _capi as capi
    from pyamg_amd.aggregation import _device_transpose, device_products
    rng = np.random.default_rng(17)
    for (nbr, nbc, R, Cc, dens) in ((300, 200, 1, 1, 0.05), (150, 220, 3, 2, 0.04), (64, 64, 2, 2, 0.2)):
        pat = _shuffle_rows(sp.random_array((nbr, nbc), density=dens, random_state=rng, format="csr"), rng)
        data = rng.standard_normal((pat.nnz, R, Cc))
        A = sp.bsr_array((data, pat.indices.astype(np.int32), pat.indptr.astype(np.int32)), shape=(nbr * R, nbc * Cc))
        ref = A.T
        Bp = np.empty(nbc + 1, dtype=np.int32)
        Bi = np.empty(pat.nnz, dtype=np.int32)
        Bx = np.empty((pat.nnz, Cc, R))
        capi.check(capi.lib().pamg_bsr_transpose_f64(nbr, nbc, R, Cc, capi.ptr(A.indptr), capi.ptr(A.indices), capi.ptr(np.ascontiguousarray(A.data).ravel()),
                                                     capi.ptr(Bp), capi.ptr(Bi), capi.ptr(Bx)), "pamg_bsr_transpose")
        assert np.array_equal(Bp, ref.indptr) and np.array_equal(Bi, ref.indices) and np.array_equal(Bx, ref.data)
    n = 1200
    big = sp.bsr_array(sp.random_array((n, n), density=0.9, random_state=rng, format="csr"), blocksize=(1, 1))
    big.indices, big.indptr = big.indices.astype(np.int32), big.indptr.astype(np.int32)
    assert big.nnz >= (1 << 20)
    want = big.T
    with device_products():
        got = big.T
        small = A.T
    assert got.format == "bsr" and np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.array_equal(got.data, want.data) and np.array_equal(small.data, ref.data)
    assert np.array_equal(_device_transpose(big).data, want.data)
    assert "transpose" not in sp.bsr_array.__dict__          # the hook is gone again


def test_symmetric_strength_against_the_reference():
    pyamg = _reference()
    from pyamg.strength import symmetric_strength_of_connection as ref_soc
    from pyamg_amd.aggregation import symmetric_strength_of_connection
    rng = np.random.default_rng(12)
    A = pyamg.gallery.poisson((30, 30, 30), format="csr")
    B = sp.random_array((500, 500), density=0.02, random_state=rng, format="csr")
    B = _shuffle_rows((B + B.T + sp.diags_array(rng.standard_normal(500))).tocsr(), rng)      # unsorted rows, signed diagonal
    Z = sp.csr_array((B.data.copy(), B.indices.copy(), B.indptr.copy()), shape=B.shape)
    Z.data[Z.indices == np.repeat(np.arange(500), np.diff(Z.indptr))] = 0.0                  # zero diagonals
    for M in (A, B, Z, sp.csr_array((7, 7))):
        for theta in (0, 0.0, 0.25, 0.9):
            Sr, Sd = ref_soc(M.copy(), theta), symmetric_strength_of_connection(M.copy(), theta)
            _same(Sd, Sr)
    # BSR: block pattern (theta = 0) / block Frobenius norms
    Ab = sp.bsr_array(sp.kron(B, np.array([[2.0, 0.5], [0.5, 3.0]])), blocksize=(2, 2))
    for theta in (0, 0.3):
        _same(symmetric_strength_of_connection(Ab, theta), ref_soc(Ab, theta))
    with pytest.raises(ValueError):
        symmetric_strength_of_connection(A, -1.0)
    with pytest.raises(TypeError):
        symmetric_strength_of_connection(A.tocsc())


def test_device_products_hook_returns_scipys_arrays():
    """sparse @ sparse inside device_products(): the same arrays as outside, other operand kinds untouched"""
    from pyamg_amd.aggregation import device_products
    rng = np.random.default_rng(13)
    A = _shuffle_rows(sp.random_array((300, 300), density=0.03, random_state=rng, format="csr"), rng)
    P = sp.random_array((300, 40), density=0.1, random_state=rng, format="csr")
    Pb, Rb = sp.bsr_array(P, blocksize=(1, 1)), sp.bsr_array(P.T.tocsr(), blocksize=(1, 1))
    x = rng.standard_normal(300)
    ref = [A @ A, Rb @ A @ Pb, Rb @ sp.bsr_array(A, blocksize=(1, 1)), A @ P.tocsc(), A @ x, sp.csr_matrix(A) @ sp.csr_matrix(P)]
    with device_products():
        assert "_matmul_sparse" in sp.csr_array.__dict__
        got = [A @ A, Rb @ A @ Pb, Rb @ sp.bsr_array(A, blocksize=(1, 1)), A @ P.tocsc(), A @ x, sp.csr_matrix(A) @ sp.csr_matrix(P)]
    assert "_matmul_sparse" not in sp.csr_array.__dict__
    for g, r in zip(got, ref):
        if sp.issparse(r):
            assert type(g) is type(r)
            _same(g, r)
        else:
            assert np.array_equal(g, r)


def test_device_setup_inside_the_reference_solver():
    """smoothed_aggregation_solver with the setup pieces patched in: same level sizes, operators within 1e-12, and the
    device cycle on the resulting hierarchy converges like the reference's own."""
    pyamg = _reference()
    from pyamg_amd import DeviceMultilevelSolver
    from pyamg_amd.aggregation import device_setup, galerkin_product
    A = pyamg.gallery.poisson((40, 40, 40), format="csr")
    np.random.seed(9)
    ml_ref = pyamg.smoothed_aggregation_solver(A.copy(), max_coarse=10)
    before = pyamg.aggregation.aggregation.jacobi_prolongation_smoother
    np.random.seed(9)
    with device_setup(pyamg):
        assert pyamg.aggregation.aggregation.jacobi_prolongation_smoother is not before
        ml_dev = pyamg.smoothed_aggregation_solver(A.copy(), max_coarse=10)
    assert pyamg.aggregation.aggregation.jacobi_prolongation_smoother is before
    assert len(ml_dev.levels) == len(ml_ref.levels)
    for k, (Ld, Lr) in enumerate(zip(ml_dev.levels, ml_ref.levels)):
        assert Ld.A.shape == Lr.A.shape and Ld.A.nnz == Lr.A.nnz
        assert np.array_equal(Ld.A.indptr, Lr.A.indptr) and np.array_equal(Ld.A.indices, Lr.A.indices)      # stored order too
        d = abs(sp.csr_array(Ld.A) - sp.csr_array(Lr.A))
        # the spectral radius' last bits travel down: 1e-14 on level 1, looser on the tiny coarse levels
        assert d.max() <= (1e-14 if k <= 1 else 1e-10) * abs(Lr.A).max(), (k, d.max())
    # the Galerkin product of the reference's own operators, level by level
    for Lr, Ln in zip(ml_ref.levels[:-1], ml_ref.levels[1:]):
        _same(galerkin_product(Lr.R, Lr.A, Lr.P), Lr.R @ Lr.A @ Lr.P)
    b = np.random.rand(A.shape[0])
    r_ref, r_dev = [], []
    ml_ref.solve(b, tol=1e-10, residuals=r_ref)
    DeviceMultilevelSolver(ml_dev).solve(b, tol=1e-10, residuals=r_dev)
    assert len(r_ref) == len(r_dev)
    assert np.max(np.abs(np.array(r_dev) - np.array(r_ref))) <= 1e-8 * r_ref[0]
