"""GPU parity, kernel level: HIP kernels (through the C ABI) vs outputs of the real
reference committed in tests/golden/kernels.npz and vs the oracle on seeded inputs.
Bar: BIT-EXACT for every kernel (f64 and f32) -- the engine reproduces the reference's
summation order with unfused multiply/add."""
import numpy as np
import pytest
import scipy.sparse as sp

from pyamg_amd import _capi as capi
from pyamg_amd import amg_core as gcore
from pyamg_amd import relaxation as grelax
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix

pytestmark = pytest.mark.gpu


def _csr(z, tag):
    n = z[f"{tag}.indptr"].size - 1
    return sp.csr_array((z[f"{tag}.data"], z[f"{tag}.indices"], z[f"{tag}.indptr"]), shape=(n, n))


def _dev(*arrs):
    return [capi.DeviceArray.from_host(a) for a in arrs]


@pytest.mark.parametrize("tag", ["pois", "irr"])
@pytest.mark.parametrize("npl", [2])
@pytest.mark.parametrize("cap", [64, 2048])
def test_spmv_family_bit_exact(kernels_npz, tag, npl, cap):
    z = kernels_npz
    A = _csr(z, tag)
    x, b = z[f"{tag}.x"], z[f"{tag}.b"]
    dA = DeviceMatrix(sparse_op(A))
    dA.tune(lds_entries=cap, nnz_per_lane=npl)
    dx, db = _dev(x, b)
    dy = capi.DeviceArray(A.shape[0], np.float64)
    dA.spmv(capi.SPMV_SET, dx, dy)
    assert np.array_equal(dy.download(), z[f"{tag}.Ax"])
    dA.spmv(capi.SPMV_RESID, dx, dy, b=db)
    assert np.array_equal(dy.download(), b - A @ x)
    dy.upload(b)
    dA.spmv(capi.SPMV_ACC, dx, dy)
    assert np.array_equal(dy.download(), b + A @ x)
    dA.spmv(capi.SPMV_AXPBY, dx, dy, b=db, c=0.37)
    assert np.array_equal(dy.download(), 0.37 * b + A @ x)
    dy.upload(x)
    dA.spmv(capi.SPMV_ACC_AXPBY, dx, dy, b=db, c=-1.7)
    assert np.array_equal(dy.download(), x + (-1.7 * b + A @ x))
    out = capi.DeviceArray(1, np.float64)
    dA.resid_sumsq(dx, db, out)
    r = b - A @ x
    assert np.isclose(out.download()[0], np.dot(r, r), rtol=1e-13)


@pytest.mark.parametrize("tag", ["pois", "irr"])
@pytest.mark.parametrize("npl,cap", [(2, 64), (2, 2048)])
def test_relaxation_bit_exact_vs_reference_outputs(kernels_npz, tag, npl, cap):
    z = kernels_npz
    A = _csr(z, tag)
    x, b = z[f"{tag}.x"], z[f"{tag}.b"]
    dA = DeviceMatrix(sparse_op(A))
    dA.tune(lds_entries=cap, nnz_per_lane=npl)
    (db,) = _dev(b)
    work = capi.DeviceArray(3 * A.shape[0], np.float64)
    for sweep in ("forward", "backward", "symmetric"):
        (dx,) = _dev(x)
        dA.gauss_seidel(dx, db, sweep=sweep, iterations=2)
        assert np.array_equal(dx.download(), z[f"{tag}.gs.{sweep}"]), sweep
        (dx,) = _dev(x)
        dA.gauss_seidel(dx, db, sweep=sweep, omega=1.3, iterations=2)      # == relaxation.sor
        assert np.array_equal(dx.download(), z[f"{tag}.sor.{sweep}"]), sweep
    (dx,) = _dev(x)
    dA.jacobi(dx, db, work, 0.8, iterations=3)
    assert np.array_equal(dx.download(), z[f"{tag}.jacobi"])
    (dx,) = _dev(x)
    dA.polynomial(dx, db, work, [0.05, -0.4, 0.9], iterations=2)
    assert np.array_equal(dx.download(), z[f"{tag}.poly"])
    (dx,) = _dev(np.zeros_like(x))
    dA.polynomial(dx, db, work, [0.05, -0.4, 0.9], iterations=1, x_is_zero=True)
    assert np.array_equal(dx.download(), z[f"{tag}.poly0"])
    # BSR(1,1) flavour: arithmetic order of amg_core::bsr_jacobi / bsr_gauss_seidel
    dB = DeviceMatrix(sparse_op(A.tobsr(blocksize=(1, 1))))
    dB.tune(lds_entries=cap, nnz_per_lane=npl)
    (dx,) = _dev(x)
    dB.jacobi(dx, db, work, 0.8, iterations=2)
    assert np.array_equal(dx.download(), z[f"{tag}.bsr1.jacobi"])
    (dx,) = _dev(x)
    dB.gauss_seidel(dx, db, sweep="symmetric", iterations=1)
    assert np.array_equal(dx.download(), z[f"{tag}.bsr1.gs"])


def test_block_kernels_bit_exact(kernels_npz):
    z = kernels_npz
    nb = z["el.indptr"].size - 1
    E = sp.bsr_array((z["el.data"].reshape(-1, 2, 2), z["el.indices"], z["el.indptr"]), shape=(2 * nb, 2 * nb))
    x, b, Dinv = z["el.x"], z["el.b"], z["el.Dinv"]
    dE = DeviceMatrix(sparse_op(E))
    dx, db = _dev(x, b)
    dy = capi.DeviceArray(E.shape[0], np.float64)
    dE.spmv(capi.SPMV_SET, dx, dy)
    assert np.array_equal(dy.download(), z["el.Ax"])
    work = capi.DeviceArray(E.shape[0], np.float64)
    dE.jacobi(dx, db, work, 0.6, iterations=2)
    assert np.array_equal(dx.download(), z["el.jacobi"])
    (dx,) = _dev(x)
    dE.gauss_seidel(dx, db, sweep="symmetric", iterations=1)
    assert np.array_equal(dx.download(), z["el.gs"])
    (dD,) = _dev(Dinv.reshape(-1))
    (dx,) = _dev(x)
    dE.block_jacobi(dx, db, work, dD, 0.7, iterations=2)
    assert np.array_equal(dx.download(), z["el.bjacobi"])
    (dx,) = _dev(x)
    dE.block_gauss_seidel(dx, db, dD, sweep="symmetric", iterations=1)
    assert np.array_equal(dx.download(), z["el.bgs"])
    # non-square blocks (prolongator of an elasticity hierarchy)
    m = z["P.meta"]
    P = sp.bsr_array((z["P.data"].reshape(-1, m[2], m[3]), z["P.indices"], z["P.indptr"]), shape=(m[0], m[1]))
    dP = DeviceMatrix(sparse_op(P))
    (dxc,) = _dev(z["P.x"])
    dy = capi.DeviceArray(P.shape[0], np.float64)
    dP.spmv(capi.SPMV_SET, dxc, dy)
    assert np.array_equal(dy.download(), z["P.Ax"])


def test_layer1_amg_core_signatures(kernels_npz):
    """Layer 1 (host buffers, reference argument order) vs the oracle, incl. partial sweeps."""
    from oracle import oracle as orc
    z = kernels_npz
    A = _csr(z, "irr")
    n = A.shape[0]
    x, b = z["irr.x"], z["irr.b"]
    Ap, Aj, Ax = A.indptr, A.indices, A.data
    y = b.copy(); y2 = b.copy()
    gcore.csr_matvec(n, n, Ap, Aj, Ax, x, y)
    orc.csr_matvec(n, Ap, Aj, Ax, x, y2)
    assert np.array_equal(y, y2)
    for (r0, r1, rs) in [(0, n, 1), (n - 1, -1, -1), (10, 200, 1), (250, 30, -1), (3, 299, 2), (298, 2, -4)]:
        a = x.copy(); c = x.copy()
        gcore.gauss_seidel(Ap, Aj, Ax, a, b, r0, r1, rs)
        orc.gauss_seidel(Ap, Aj, Ax, c, b, r0, r1, rs)
        assert np.array_equal(a, c), (r0, r1, rs)
        a = x.copy(); c = x.copy()
        gcore.sor_gauss_seidel(Ap, Aj, Ax, a, b, r0, r1, rs, 0.9)
        orc.sor_gauss_seidel(Ap, Aj, Ax, c, b, r0, r1, rs, 0.9)
        assert np.array_equal(a, c), (r0, r1, rs)
        a = x.copy(); c = x.copy(); t1 = np.full(n, -7.0); t2 = np.full(n, -7.0)
        gcore.jacobi(Ap, Aj, Ax, a, b, t1, r0, r1, rs, np.array([0.7]))
        orc.jacobi(Ap, Aj, Ax, c, b, t2, r0, r1, rs, 0.7)
        assert np.array_equal(a, c) and np.array_equal(t1, t2), (r0, r1, rs)
    a = x.copy(); c = x.copy()
    gcore.bsr_gauss_seidel(Ap, Aj, Ax, a, b, 0, n, 1, 1)
    orc.bsr_gauss_seidel(Ap, Aj, Ax, c, b, 0, n, 1, 1)
    assert np.array_equal(a, c)
    a = x.copy(); c = x.copy(); t1 = np.empty(n); t2 = np.empty(n)
    gcore.bsr_jacobi(Ap, Aj, Ax, a, b, t1, 0, n, 1, 1, np.array([0.7]))
    orc.bsr_jacobi(Ap, Aj, Ax, c, b, t2, 0, n, 1, 1, 0.7)
    assert np.array_equal(a, c)
    # blocks
    nb = z["el.indptr"].size - 1
    Ep, Ej, Ex = z["el.indptr"], z["el.indices"], z["el.data"]
    xe, be, Dinv = z["el.x"], z["el.b"], np.ascontiguousarray(z["el.Dinv"]).reshape(-1)
    ye = be.copy(); ye2 = be.copy()
    gcore.bsr_matvec(nb, nb, 2, 2, Ep, Ej, Ex, xe, ye)
    orc.bsr_matvec(nb, 2, 2, Ep, Ej, Ex, xe, ye2)
    assert np.array_equal(ye, ye2)
    for (r0, r1, rs) in [(0, nb, 1), (nb - 1, -1, -1), (5, 40, 1)]:
        a = xe.copy(); c = xe.copy()
        gcore.bsr_gauss_seidel(Ep, Ej, Ex, a, be, r0, r1, rs, 2)
        orc.bsr_gauss_seidel(Ep, Ej, Ex, c, be, r0, r1, rs, 2)
        assert np.array_equal(a, c), (r0, r1, rs)
        a = xe.copy(); c = xe.copy()
        gcore.block_gauss_seidel(Ep, Ej, Ex, a, be, Dinv, r0, r1, rs, 2)
        orc.block_gauss_seidel(Ep, Ej, Ex, c, be, Dinv, r0, r1, rs, 2)
        assert np.array_equal(a, c), (r0, r1, rs)
    a = xe.copy(); c = xe.copy(); t1 = np.empty_like(xe); t2 = np.empty_like(xe)
    gcore.bsr_jacobi(Ep, Ej, Ex, a, be, t1, 0, nb, 1, 2, np.array([0.6]))
    orc.bsr_jacobi(Ep, Ej, Ex, c, be, t2, 0, nb, 1, 2, 0.6)
    assert np.array_equal(a, c)
    a = xe.copy(); c = xe.copy()
    gcore.block_jacobi(Ep, Ej, Ex, a, be, Dinv, t1, 0, nb, 1, np.array([0.7]), 2)
    orc.block_jacobi(Ep, Ej, Ex, c, be, Dinv, t2, 0, nb, 1, 0.7, 2)
    assert np.array_equal(a, c)
    with pytest.raises(TypeError):          # .noconvert() semantics of the reference bindings
        gcore.gauss_seidel(Ap, Aj, Ax, x.astype(np.float32), b, 0, n, 1)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_edge_cases_and_long_rows(dtype):
    """Empty operator, empty rows, one row longer than the LDS window, n = 1; vs the oracle."""
    from oracle import oracle as orc
    rng = np.random.RandomState(11)
    # long rows: dense-ish 40 x 3000 block + identity rows, cap far below the row length
    n = 3000
    M = sp.random(n, n, density=0.002, random_state=rng, format="lil")
    M[7, :] = rng.rand(n)                    # 3000 entries in one row
    M[8, ::2] = rng.rand(n // 2)
    M.setdiag(rng.rand(n) + 2.0)
    M[20, :] = 0                             # empty row
    M = sp.csr_array(M.tocsr()).astype(dtype)
    M.sort_indices()
    op = sparse_op(M)
    x = rng.rand(n).astype(dtype); b = rng.rand(n).astype(dtype)
    for npl, cap in [(2, 64), (2, 128), (2, 2048)]:
        dM = DeviceMatrix(op)
        dM.tune(lds_entries=cap, nnz_per_lane=npl)
        dx, db = _dev(x, b)
        dy = capi.DeviceArray(n, dtype)
        dM.spmv(capi.SPMV_SET, dx, dy)
        assert np.array_equal(dy.download(), orc.matvec(op, x)), (npl, cap)
        c = x.copy()
        orc.relax_gauss_seidel(op, c, b, 1, "symmetric")
        dM.gauss_seidel(dx, db, sweep="symmetric")
        assert np.array_equal(dx.download(), c), (npl, cap)
        c = x.copy()
        orc.relax_jacobi(op, c, b, 2, 0.5)
        (dx,) = _dev(x)
        work = capi.DeviceArray(n, dtype)
        dM.jacobi(dx, db, work, 0.5, iterations=2)
        assert np.array_equal(dx.download(), c), (npl, cap)
    # n = 1 and an all-zero operator
    one = sparse_op(sp.csr_array(np.array([[2.0]], dtype=dtype)))
    dO = DeviceMatrix(one)
    dx, db = _dev(np.array([0.0], dtype=dtype), np.array([10.0], dtype=dtype))
    dO.gauss_seidel(dx, db)
    assert dx.download()[0] == 5.0
    Z = sparse_op(sp.csr_array((5, 5), dtype=dtype))
    dZ = DeviceMatrix(Z)
    dx, db = _dev(np.arange(5, dtype=dtype), np.ones(5, dtype=dtype))
    dy = capi.DeviceArray(5, dtype)
    dZ.spmv(capi.SPMV_RESID, dx, dy, b=db)
    assert np.array_equal(dy.download(), np.ones(5, dtype=dtype))
    dZ.gauss_seidel(dx, db)                  # zero diagonal everywhere: x untouched
    assert np.array_equal(dx.download(), np.arange(5, dtype=dtype))
    E0 = sparse_op(sp.csr_array((0, 0), dtype=dtype))
    dE = DeviceMatrix(E0)
    assert dE.info()["rows"] == 0


def test_relaxation_module_contract():
    """The NumPy-facing twins keep the reference's error contract (test_relaxation.py:48-111)
    and known answers (:148-197, :299-346, :808-835)."""
    import json
    from conftest import GOLDEN
    ka = json.loads((GOLDEN / "known_answers.json").read_text())

    def tri(N):
        return sp.diags_array([2 * np.ones(N), -np.ones(N), -np.ones(N)], offsets=[0, -1, 1], shape=(N, N), format="csr")
    for c in ka["jacobi"]["cases"]:
        x = np.array(c["x"]); grelax.jacobi(tri(c["N"]), x, np.array(c["b"]), omega=c["omega"])
        np.testing.assert_almost_equal(x, c["expect"])
    for c in ka["gauss_seidel"]["cases"]:
        x = np.array(c["x"]); grelax.gauss_seidel(tri(c["N"]), x, np.array(c["b"]), sweep=c["sweep"])
        np.testing.assert_almost_equal(x, c["expect"])
    s = ka["sor_wikipedia"]
    for its, exp in s["expect_after"].items():
        x = np.zeros(4)
        grelax.sor(sp.csr_array(np.array(s["A"])), x, np.array(s["b"]), s["omega"], iterations=int(its))
        np.testing.assert_allclose(x, exp, rtol=s["rtol"])
    A = tri(10)
    x = np.zeros(10); b = np.ones(10)
    with pytest.raises(TypeError):
        grelax.jacobi(A, x.astype(np.float32), b)
    with pytest.raises(ValueError):
        grelax.gauss_seidel(A, np.zeros(20)[::2], b)
    with pytest.raises(ValueError):
        grelax.gauss_seidel(A, x, np.ones(11))
    with pytest.raises(ValueError):
        grelax.gauss_seidel(A, x, b, sweep="sideways")
    with pytest.raises(ValueError):
        grelax.jacobi(sp.csr_array(np.ones((3, 4))), np.zeros(3), np.zeros(3))
    # x given as (n,1): updated in place like the reference
    x2 = np.zeros((10, 1)); grelax.gauss_seidel(A, x2, np.ones((10, 1)))
    x1 = np.zeros(10); grelax.gauss_seidel(A, x1, b)
    assert np.array_equal(x2.ravel(), x1) and x1.any()


def test_gs_sweep_modes_all_exact():
    """Every scheduler of the order-exact sweep (one launch per level, one persistent workgroup,
    granular sync-free sweep across the chip / inside one XCD / on a tiny grid, automatic choice)
    gives the reference's bits -- on a stencil, an irregular symmetric pattern, a BSR(1,1)
    operator, a NON-symmetric pattern (old values from a snapshot), an operator with zero and
    missing diagonal entries, and in single precision."""
    from oracle import oracle as orc
    from tools.problems import poisson_csr
    rng = np.random.RandomState(3)
    A3 = poisson_csr((24, 20, 22))
    S = sp.random(4000, 4000, density=0.004, random_state=rng, format="csr")
    S = sp.csr_array(S + S.T + sp.diags_array(rng.rand(4000) + 4.0))
    S.sort_indices()
    N = sp.random(3000, 3000, density=0.004, random_state=rng, format="csr")
    N = sp.csr_array(N + sp.diags_array(rng.rand(3000) + 4.0))          # structurally non-symmetric
    Z = sp.lil_array(S[:1500, :1500])
    for i in range(0, 1500, 7):
        Z[i, i] = 0.0                                                    # explicit zero / missing diagonals
    Z = sp.csr_array(Z)
    Z.sort_indices()
    cases = [sparse_op(A3), sparse_op(S), sparse_op(A3.tobsr(blocksize=(1, 1))), sparse_op(N), sparse_op(Z),
             sparse_op(sp.csr_array(S.astype(np.float32)))]
    for op in cases:
        n = op.shape[0]
        dt = op.data.dtype
        x = rng.rand(n).astype(dt); b = rng.rand(n).astype(dt)
        ref = x.copy(); orc.relax_gauss_seidel(op, ref, b, 2, "symmetric")
        refs = x.copy(); orc.relax_sor(op, refs, b, 1.4, 1, "forward")
        dA = DeviceMatrix(op)
        dA.tune(lds_entries=256)
        db = capi.DeviceArray.from_host(b)
        for kw in (dict(gs_mode=1), dict(gs_mode=3), dict(gs_mode=2, gran_xcd=2, gran_cap=0),
                   dict(gs_mode=2, gran_xcd=1, gran_cap=0), dict(gs_mode=2, gran_xcd=2, gran_cap=3),
                   dict(gs_mode=2, gran_xcd=1, gran_cap=2), dict(gs_mode=0, gran_xcd=0, gran_cap=0),
                   # tiled sweep: automatic tiling; one tile; many small tiles with a tiny ring (in-tile values that
                   # fall out of the ring go through the global hand-off) and short steps; the narrow-step kernel
                   dict(gs_mode=5, tile_G=0, tile_W=2048, tile_cap=0), dict(gs_mode=5, tile_G=1),
                   dict(gs_mode=5, tile_G=37, tile_W=256, tile_cap=96), dict(gs_mode=5, tile_G=200, tile_W=1024, tile_cap=900)):
            dA.tune(**kw)
            dx = capi.DeviceArray.from_host(x)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            assert np.array_equal(dx.download(), ref), (kw, op.fmt, dt)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="forward", omega=1.4)
            assert np.array_equal(dx.download(), refs), (kw, op.fmt, dt)
            assert not dA.flow_error(), kw


def test_gs_merged_fast_order_agrees_to_rounding():
    """The MERGED fast order (round 6; tune lane_merge = s, csrc/pamg_lanem_plan.h, gs_lanem_kernel in pamg_lane.hip): s consecutive dependency
    levels of the reference's sweep (relaxation.h:48-76) are eliminated algebraically into one super-level, the sweep pays one hand-off per
    super-level.  Against the ORACLE's sequential sweep (1e-13 relative per call) and the order-exact device sweep, s = 2 .. 8, static form across the
    chip and ticket form inside one XCD, one row per wave and TWO (32 lanes each, rows of a super-level paired by length), with and without the gate
    operand, tiny grids (waves that wait), forward / backward / symmetric;
    rows with zero / missing diagonals stay untouched; a structurally non-symmetric pattern; bit-reproducible; an operator that is NOT
    diagonally dominant must make the planner close groups early (growth bound) or decline, and still give the sequential sweep's answer; SOR on
    the same operator takes the unmerged layout; f32 keeps the unmerged form."""
    from oracle import oracle as orc
    rng = np.random.RandomState(11)

    def sa_like(n, density, seed):
        r = np.random.RandomState(seed)
        S = sp.random(n, n, density=density, random_state=r, format="csr")
        S = sp.csr_array(-abs(S + S.T))
        S.setdiag(0)
        S.eliminate_zeros()
        d = np.asarray(abs(S).sum(axis=1)).ravel() + 0.5 + r.rand(n)
        A = sp.csr_array(S + sp.diags_array(d))
        A.sort_indices()
        return A

    S = sa_like(6000, 0.003, 1)                                              # ~36 entries per row
    Z = sp.lil_array(sa_like(2500, 0.008, 2))
    for i in range(0, 2500, 7):
        Z[i, i] = 0.0
    Z = sp.csr_array(Z)
    N = sp.random(3000, 3000, density=0.006, random_state=rng, format="csr")
    N = sp.csr_array(N + sp.diags_array(rng.rand(3000) + 6.0))                 # structurally non-symmetric
    n_nd = 3000
    main, off = np.full(n_nd, 1.0), np.full(n_nd - 1, -3.0)
    off[23::24] = 0.0
    ND = sp.csr_array(sp.diags_array([off, main, 0.1 * off], offsets=[-1, 0, 1]))   # not diagonally dominant: independent chains of 24 rows with ratio 3 (3^23 per sweep: finite)
    for ci, M in enumerate((S, Z, N, ND)):
        op = sparse_op(M)
        n = op.shape[0]
        x, b = rng.rand(n), rng.rand(n)
        ref = x.copy(); orc.relax_gauss_seidel(op, ref, b, 2, "symmetric")
        reff = x.copy(); orc.relax_gauss_seidel(op, reff, b, 1, "forward")
        refb = x.copy(); orc.relax_gauss_seidel(op, refb, b, 1, "backward")
        refs = x.copy(); orc.relax_sor(op, refs, b, 1.3, 1, "forward")
        tol = 1e-13 if ci != 3 else 1e-10                                    # growth factors up to the cap (1e3) cost up to three digits
        dA = DeviceMatrix(op)
        db, dx = capi.DeviceArray.from_host(b), capi.DeviceArray.from_host(x)
        dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
        assert np.array_equal(dx.download(), ref)                           # a bare operator is order-exact
        dA.tune(gs_order=1, lane_wide=1, line_scan=0)
        hops = {}
        for kw in (dict(lane_merge=1), dict(lane_merge=2), dict(lane_merge=3), dict(lane_merge=3, gran_xcd=1), dict(lane_merge=4, gran_xcd=2, lane_G=3),
                   dict(lane_merge=8, gran_xcd=1, lane_G=1), dict(lane_merge=5, gran_xcd=0, lane_G=0, lane_flags=0), dict(lane_merge=0, lane_flags=1, lanem_ahead=60),
                   dict(lane_merge=3, lanem_rpw=2, gran_xcd=2), dict(lane_merge=2, lanem_rpw=2, gran_xcd=1, lane_G=2), dict(lane_merge=6, lanem_rpw=2, gran_xcd=0, lane_flags=0),
                   dict(lane_merge=3, lanem_rpw=0)):
            dA.tune(**kw)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            got = dx.download()
            mi = dA.lanem_info(0)
            if kw["lane_merge"] == 1:
                assert mi["rows"] == 0 and dA.lane_info(0)["groups"] > 0
            elif ci != 3:
                assert mi["rows"] == n and mi["super_levels"] < mi["dependency_levels"], (kw, ci, mi)   # the merged form really ran
                assert mi["max_growth"] < 50.0
            else:
                assert mi["rows"] == 0 or mi["max_growth"] <= 1e3, (kw, mi)    # the growth bound holds for whatever was merged
                if kw["lane_merge"] == 8:
                    assert mi["rows"] == 0 or mi["closed_by_growth"] > 0, (kw, mi)   # 3^7 > 1e3: groups closed early (or the form declined)
            hops[kw["lane_merge"]] = mi["super_levels"]
            assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)), (kw, ci, np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            assert np.array_equal(dx.download(), got), (kw, ci)              # the order of the additions is the layout's, never the timing's
            for sw, r_ in (("forward", reff), ("backward", refb)):
                dx.upload(x)
                dA.gauss_seidel(dx, db, sweep=sw)
                assert np.max(np.abs(dx.download() - r_)) <= tol * np.max(np.abs(r_)), (kw, ci, sw)
            if ci == 1:
                assert np.array_equal(got[0::7], x[0::7])                   # zero diagonals: untouched
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="forward", omega=1.3)              # SOR: the unmerged layout of the same schedule
            assert np.max(np.abs(dx.download() - refs)) <= tol * np.max(np.abs(refs)), (kw, ci)
            assert not dA.flow_error(), kw
        if ci == 0:
            assert hops[2] <= (mi["dependency_levels"] + 1) // 2 + 2 and hops[3] < hops[2]
        dA.tune(gs_order=0)
        dx.upload(x)
        dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
        assert np.array_equal(dx.download(), ref)                           # and back to the reference's bits
        dA.free(); dx.free(); db.free()
    # f32: the merged form is f64-only, the unmerged lane form runs
    op32 = sparse_op(sp.csr_array(S.astype(np.float32)))
    dA = DeviceMatrix(op32)
    dA.tune(gs_order=1, lane_wide=1, line_scan=0, lane_merge=3)
    x32, b32 = rng.rand(op32.shape[0]).astype(np.float32), rng.rand(op32.shape[0]).astype(np.float32)
    r32 = x32.copy(); orc.relax_gauss_seidel(op32, r32, b32, 1, "symmetric")
    dx, db = capi.DeviceArray.from_host(x32), capi.DeviceArray.from_host(b32)
    dA.gauss_seidel(dx, db, sweep="symmetric")
    assert dA.lanem_info(0)["rows"] == 0 and dA.lane_info(0)["groups"] > 0
    assert np.max(np.abs(dx.download() - r32)) <= 2e-6 * np.max(np.abs(r32))
    dA.free(); dx.free(); db.free()


def test_gs_fast_order_agrees_to_rounding():
    """Fast order (tune gs_order=1, pamg_lane.hip): same sweep order over the rows, lane-parallel row sums and
    multiplication by 1/a_ii -- every form (automatic lane width, forced widths, static assignment across the chip,
    ticket form inside one XCD, tiny grids) must agree with the order-exact device sweep (= the reference's bits,
    relaxation.h:48-76,116-145,185-266) to 1e-13 relative per sweep (f32: 2e-6), on a stencil, an irregular symmetric
    pattern, BSR(1,1), a NON-symmetric pattern (snapshot), zero / missing diagonals, and in single precision; exact mode
    is untouched by the switch."""
    from oracle import oracle as orc
    from tools.problems import poisson_csr
    rng = np.random.RandomState(3)
    A3 = poisson_csr((24, 20, 22))
    S = sp.random(4000, 4000, density=0.004, random_state=rng, format="csr")
    S = sp.csr_array(S + S.T + sp.diags_array(rng.rand(4000) + 4.0))
    S.sort_indices()
    N = sp.random(3000, 3000, density=0.004, random_state=rng, format="csr")
    N = sp.csr_array(N + sp.diags_array(rng.rand(3000) + 4.0))          # structurally non-symmetric
    Z = sp.lil_array(S[:1500, :1500])
    for i in range(0, 1500, 7):
        Z[i, i] = 0.0                                                    # explicit zero / missing diagonals
    Z = sp.csr_array(Z)
    Z.sort_indices()
    D = sp.random(1200, 1200, density=0.06, random_state=rng, format="csr")   # ~70 entries per row: 32 lanes per row
    D = sp.csr_array(D + D.T + sp.diags_array(rng.rand(1200) + 80.0))
    D.sort_indices()
    cases = [sparse_op(A3), sparse_op(S), sparse_op(A3.tobsr(blocksize=(1, 1))), sparse_op(N), sparse_op(Z), sparse_op(D),
             sparse_op(sp.csr_array(S.astype(np.float32)))]
    for ci, op in enumerate(cases):
        n = op.shape[0]
        dt = op.data.dtype
        tol = 1e-13 if dt == np.float64 else 2e-6
        x = rng.rand(n).astype(dt); b = rng.rand(n).astype(dt)
        ref = x.copy(); orc.relax_gauss_seidel(op, ref, b, 2, "symmetric")
        refs = x.copy(); orc.relax_sor(op, refs, b, 1.4, 1, "forward")
        dA = DeviceMatrix(op)
        db = capi.DeviceArray.from_host(b)
        dx = capi.DeviceArray.from_host(x)
        dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
        assert np.array_equal(dx.download(), ref)                       # a bare operator is order-exact
        seen = set()
        for kw in (dict(gs_order=1, lane_wide=1, line_scan=0, lane_merge=1), dict(lane_L=16), dict(lane_L=64), dict(lane_L=0, gran_xcd=1), dict(gran_xcd=2, lane_G=3),
                   dict(gran_xcd=1, lane_G=1), dict(gran_xcd=0, lane_G=0, lane_L=8), dict(lane_flags=0), dict(lane_flags=1, gran_xcd=2)):
            dA.tune(**kw)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            got = dx.download()
            info = dA.lane_info(0)
            assert info["groups"] > 0, (kw, ci)                         # the lane form really ran
            seen.add(info["lanes_per_row"])
            assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)), (kw, ci, np.max(np.abs(got - ref)))
            # the order of the additions is the layout's, never the timing's: a second run gives the same bits
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            assert np.array_equal(dx.download(), got), (kw, ci)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="forward", omega=1.4)
            got = dx.download()
            assert np.max(np.abs(got - refs)) <= tol * np.max(np.abs(refs)), (kw, ci)
            assert not dA.flow_error(), kw
        assert len(seen) >= 2 or ci == 5                                # (the ~70-per-row operator needs 64 lanes whatever is asked)
        # the line-scan form (grid stencils in their natural order: consecutive rows coupled) -- default where it applies
        for kw in (dict(line_scan=1, lane_G=0), dict(lane_G=2), dict(lane_flags=0, lane_G=0)):
            dA.tune(**kw)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            got = dx.download()
            assert (dA.line_info(0)["lines"] > 0) == (ci in (0, 2)), (ci, dA.line_info(0))
            assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)), (kw, ci, np.max(np.abs(got - ref)))
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="forward", omega=1.4)
            assert np.max(np.abs(dx.download() - refs)) <= tol * np.max(np.abs(refs)), (kw, ci)
            assert not dA.flow_error(), kw
        dA.tune(lane_flags=1)
        dA.tune(gs_order=0)
        dx.upload(x)
        dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
        assert np.array_equal(dx.download(), ref)                       # and back


def test_line_layout_filled_on_the_device_is_the_host_layout(monkeypatch):
    """The line-scan layout (chunk slots, 1 / a_ii, recurrence coefficients, gates) is written by the device from the resident
    CSR arrays by default; PAMG_LINE_HOST_FILL=1 builds the same arrays on the host (the planner the CPU suite replays) and
    uploads them.  Same plan statistics and bit-identical sweeps either way: 3-D and 2-D stencils, a zero diagonal, rows with
    a duplicate predecessor entry, f64 / f32, forward / backward / SOR."""
    from tools.problems import poisson_csr
    rng = np.random.RandomState(17)
    A3 = sp.csr_array(poisson_csr((24, 20, 22)))
    A2 = sp.lil_array(poisson_csr((96, 80)))
    for i in range(5, 7000, 97):
        A2[i, i] = 0.0                                     # zero diagonals: rows left untouched
    A2 = sp.csr_array(A2.tocsr())
    dup = sp.csr_array(poisson_csr((5000,)))
    # a duplicate of the predecessor entry in some rows (unsummed duplicates are legal CSR): the second one stays a slot
    ip, ix, dv = dup.indptr.copy(), list(dup.indices), list(dup.data)
    rows = []
    for r in range(dup.shape[0]):
        ent = [(ix[p], dv[p]) for p in range(ip[r], ip[r + 1])]
        if r % 37 == 3 and r > 0:
            ent.append((r - 1, -0.25))
        rows.append(ent)
    dup = sp.csr_array((np.array([v for e in rows for _, v in e]), np.array([c for e in rows for c, _ in e], dtype=np.int32),
                        np.cumsum([0] + [len(e) for e in rows]).astype(np.int32)), shape=dup.shape)
    for M in (A3, A2, dup, sp.csr_array(A3.astype(np.float32))):
        op = sparse_op(M) if M is not dup else sparse_op(M)
        n, dt = op.shape[0], op.data.dtype
        x, b = rng.rand(n).astype(dt), rng.rand(n).astype(dt)
        res = []
        for host in ("1", "0"):
            monkeypatch.setenv("PAMG_LINE_HOST_FILL", host)
            dA = DeviceMatrix(op)
            dA.tune(gs_order=1, line_scan=1)
            dx, db = capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(b)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            a = dx.download()
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="backward", omega=1.3)
            res.append((a, dx.download(), {k: v for k, v in dA.line_info(0).items() if k != "bytes"}, dA.line_info(1)["early_entries"]))
            assert dA.line_info(0)["lines"] > 0 and not dA.flow_error()
            dA.free()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert res[0][2] == res[1][2] and res[0][3] == res[1][3], (res[0][2], res[1][2])


def test_lane_layout_filled_on_the_device_is_the_host_layout(monkeypatch):
    """The lane-parallel layout likewise: PAMG_LANE_HOST_FILL=1 builds slots / values / reciprocal diagonals on the host and uploads
    them, the default writes them on the device from the resident CSR arrays -- bit-identical sweeps and equal statistics on an
    irregular symmetric pattern (several rows per wave), a ~70-per-row operator (one row per wave, two slots per lane), zero / missing
    diagonals, a non-symmetric pattern and f32."""
    rng = np.random.RandomState(19)
    S = sp.random(4000, 4000, density=0.004, random_state=rng, format="csr")
    S = sp.csr_array(S + S.T + sp.diags_array(rng.rand(4000) + 4.0))
    S.sort_indices()
    Z = sp.lil_array(S[:1500, :1500])
    for i in range(0, 1500, 7):
        Z[i, i] = 0.0
    Z = sp.csr_array(Z)
    D = sp.random(1200, 1200, density=0.05, random_state=rng, format="csr")
    D = sp.csr_array(D + D.T + sp.diags_array(rng.rand(1200) + 150.0))
    N = sp.random(3000, 3000, density=0.004, random_state=rng, format="csr")
    N = sp.csr_array(N + sp.diags_array(rng.rand(3000) + 4.0))
    for M in (S, Z, D, N, sp.csr_array(S.astype(np.float32))):
        op = sparse_op(M)
        n, dt = op.shape[0], op.data.dtype
        x, b = rng.rand(n).astype(dt), rng.rand(n).astype(dt)
        res = []
        for host in ("1", "0"):
            monkeypatch.setenv("PAMG_LANE_HOST_FILL", host)
            dA = DeviceMatrix(op)
            dA.tune(gs_order=1, lane_wide=1, line_scan=0, lane_merge=1)             # the unmerged layout is what is filled on the device
            dx, db = capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(b)
            dA.gauss_seidel(dx, db, sweep="symmetric", iterations=2)
            a = dx.download()
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="backward", omega=1.3)
            res.append((a, dx.download(), {k: v for k, v in dA.lane_info(0).items() if k != "bytes"}))
            assert dA.lane_info(0)["groups"] > 0 and not dA.flow_error()
            dA.free()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert res[0][2] == res[1][2], (res[0][2], res[1][2])


def test_resid_sumsq_two_stage_reduction():
    from tools.problems import poisson_csr
    A = poisson_csr((400, 400))
    dA = DeviceMatrix(sparse_op(A))
    dA.tune(lds_entries=64)
    assert dA.info()["row_blocks"] > 8192
    rng = np.random.RandomState(5)
    x = rng.rand(A.shape[0]); b = rng.rand(A.shape[0])
    dx, db = _dev(x, b)
    out = capi.DeviceArray(1, np.float64)
    dA.resid_sumsq(dx, db, out)
    r = b - A @ x
    assert np.isclose(out.download()[0], np.dot(r, r), rtol=1e-13)


def test_relaxation_module_block_doctests_and_polynomial():
    """Device twins reproduce the reference's doctest values (relaxation.py:449-460, 529-541) and
    the explicit Horner expressions of test_relaxation.py:115-146."""
    T = sp.diags_array([2 * np.ones(10), -np.ones(10), -np.ones(10)], offsets=[0, -1, 1], shape=(10, 10), format="csr")
    I = sp.eye_array(10, format="csr")
    A = sp.csr_array(sp.kron(I, T) + sp.kron(T, I)); A.sort_indices()
    B = A.tobsr(blocksize=(4, 4))
    Dinv = np.zeros((25, 4, 4))
    for i in range(25):
        for p in range(B.indptr[i], B.indptr[i + 1]):
            if B.indices[p] == i:
                Dinv[i] = np.linalg.pinv(B.data[p])
    b = np.ones((100, 1))
    x = np.zeros((100, 1)); grelax.block_jacobi(A, x, b, Dinv=Dinv, blocksize=4, iterations=10, omega=1.0)
    assert f"{np.linalg.norm(b - A @ x):2.4}" == "4.665"
    x = np.zeros((100, 1)); grelax.block_gauss_seidel(A, x, b, iterations=10, blocksize=4, sweep="symmetric", Dinv=Dinv)
    assert f"{np.linalg.norm(b - A @ x):2.4}" == "0.9583"
    T3 = sp.diags_array([2 * np.ones(3), -np.ones(3), -np.ones(3)], offsets=[0, -1, 1], shape=(3, 3), format="csr")
    x0 = np.arange(3, dtype=float)
    bv = np.array([10.0, 20.0, 30.0])
    r = bv - T3 @ x0
    x = x0.copy(); grelax.polynomial(T3, x, bv, [-0.14285714, 1.0, -2.0])
    np.testing.assert_almost_equal(x, x0 - 0.14285714 * T3 @ T3 @ r + T3 @ r - 2 * r)
    x = 0 * x0; grelax.polynomial(T3, x, bv, [-0.14285714, 1.0, -2.0])
    np.testing.assert_almost_equal(x, -0.14285714 * T3 @ T3 @ bv + T3 @ bv - 2 * bv)


def test_pinv_array_and_block_diag_bit_exact():
    """amg_core.pinv_array on the device (pamg_pinv_array: one lane per block, the reference's Jacobi SVD expression
    by expression) against the reference's outputs and the oracle -- bit for bit, n = 1..6, f64 / f32, both storage
    conventions -- and get_block_diag / the Dinv=None form of the block smoothers (reference doctests,
    relaxation.py:449-460, 529-541, run UNMODIFIED: no Dinv passed)"""
    from conftest import GOLDEN
    from oracle import oracle as orc
    z = np.load(GOLDEN / "kernels_setup.npz")
    for tag in ("f64", "f32"):
        for n in range(1, 7):
            A = z[f"pinv.{tag}.{n}.in"]
            for tr in ("T", "F"):
                out = A.copy()
                gcore.pinv_array(out.ravel(), A.shape[0], n, tr)
                assert np.array_equal(out, z[f"pinv.{tag}.{n}.{tr}"]), (tag, n, tr)
                chk = A.copy()
                orc.pinv_array(chk, A.shape[0], n, tr)
                assert np.array_equal(out, chk)
    with pytest.raises(NotImplementedError):
        gcore.pinv_array(np.zeros(2 * 49), 2, 7, "T")
    Ab = sp.bsr_array((z["bd.data"], z["bd.indices"], z["bd.indptr"]), shape=tuple(z["bd.shape"]))
    assert np.array_equal(grelax.get_block_diag(Ab.copy(), 2, inv_flag=False), z["bd.blk2"])
    Ab2 = Ab.copy()
    assert np.array_equal(grelax.get_block_diag(Ab2, 2, inv_flag=True), z["bd.inv2"])
    assert Ab2.block_D_inv is grelax.get_block_diag(Ab2, 2)                      # cached like the reference caches it
    T = sp.diags_array([2 * np.ones(10), -np.ones(10), -np.ones(10)], offsets=[0, -1, 1], shape=(10, 10), format="csr")
    I = sp.eye_array(10, format="csr")
    A = sp.csr_array(sp.kron(I, T) + sp.kron(T, I)); A.sort_indices()
    assert np.array_equal(grelax.get_block_diag(A, 4), z["bd.inv4"])
    b = np.ones((100, 1))
    x = np.zeros((100, 1)); grelax.block_jacobi(A, x, b, blocksize=4, iterations=10, omega=1.0)
    assert f"{np.linalg.norm(b - A @ x):2.4}" == "4.665"
    x = np.zeros((100, 1)); grelax.block_gauss_seidel(A, x, b, iterations=10, blocksize=4, sweep="symmetric")
    assert f"{np.linalg.norm(b - A @ x):2.4}" == "0.9583"


def test_block_sweep_modes_all_exact():
    """Block/BSR-point Gauss-Seidel: every scheduler (per-level launches, one persistent workgroup,
    granular sync-free sweep on its automatic and on a tiny grid, barrier grid, automatic choice)
    gives the oracle's bits -- BSR 2x2 on a 3-D grid, a structurally NON-symmetric 3x3 block pattern
    with missing and singular diagonal blocks (old values from a snapshot), and 6x6 blocks."""
    from oracle import oracle as orc
    from tools.problems import poisson_csr
    rng = np.random.RandomState(12)
    P = poisson_csr((24, 22, 20))
    blk = np.array([[2.0, 0.3], [-0.4, 1.5]])
    M2 = sp.kron(P, blk, format="bsr")
    # non-symmetric block pattern, 3x3 blocks, some diagonal blocks missing / with a zero pivot
    nbr = 900
    Pn = sp.random(nbr, nbr, density=0.01, random_state=rng, format="lil")
    for i in range(nbr):
        if i % 11 != 5:
            Pn[i, i] = 4.0 + rng.rand()
    for j in range(100, 140):
        Pn[450, j] = 1.0                                            # one block row longer than a small LDS window
    Pn = sp.csr_array(Pn)
    Pn.sort_indices()
    dat = rng.rand(Pn.nnz, 3, 3) - 0.5
    for i in range(nbr):
        for p in range(Pn.indptr[i], Pn.indptr[i + 1]):
            if Pn.indices[p] == i:
                dat[p] += 4.0 * np.eye(3)
                if i % 13 == 2:
                    dat[p][1, 1] = 0.0                                  # zero point diagonal inside the block
    M3 = sp.bsr_array((dat, Pn.indices.astype(np.int32), Pn.indptr.astype(np.int32)), shape=(3 * nbr, 3 * nbr), blocksize=(3, 3))
    P6 = poisson_csr((7, 6, 5))
    M6 = sp.kron(P6, rng.rand(6, 6) + 6.0 * np.eye(6), format="bsr")
    # small, dense-ish block levels (the coarse levels of an elasticity hierarchy: nearly every block row its own dependency
    # level): the single-workgroup kernel with the iterate in LDS (bsr_small_kernel) -- 6x6, a structurally non-symmetric 3x3
    # pattern with missing / singular diagonal blocks, and 5x5 (run-time block size)
    def small(nb_, bs_, dens, drop):
        Pp = sp.random(nb_, nb_, density=dens, random_state=rng, format="lil")
        for i in range(nb_):
            if not (drop and i % 9 == 4):
                Pp[i, i] = 1.0
        Pp = sp.csr_array(Pp)
        Pp.sort_indices()
        dd = rng.rand(Pp.nnz, bs_, bs_) - 0.5
        for i in range(nb_):
            for p in range(Pp.indptr[i], Pp.indptr[i + 1]):
                if Pp.indices[p] == i:
                    dd[p] += (2.0 * bs_ * dens * nb_ + 4.0) * np.eye(bs_)
                    if drop and i % 7 == 3:
                        dd[p][bs_ - 1, bs_ - 1] = 0.0
        return sp.bsr_array((dd, Pp.indices.astype(np.int32), Pp.indptr.astype(np.int32)), shape=(bs_ * nb_, bs_ * nb_), blocksize=(bs_, bs_))
    S6, S3, S5 = small(120, 6, 0.3, False), small(200, 3, 0.08, True), small(64, 5, 0.5, False)
    for M, bs in ((M2, 2), (M3, 3), (M6, 6), (S6, 6), (S3, 3), (S5, 5)):
        M = sp.bsr_array((M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32)), shape=M.shape, blocksize=(bs, bs))
        op = sparse_op(M)
        n = op.shape[0]
        nb = n // bs
        x = rng.rand(n); b = rng.rand(n)
        Dinv = np.zeros((nb, bs, bs))
        for i in range(nb):
            for p in range(M.indptr[i], M.indptr[i + 1]):
                if M.indices[p] == i:
                    Dinv[i] = np.linalg.pinv(np.asarray(M.data[p]))
        ref_pnt = x.copy(); orc.relax_gauss_seidel(op, ref_pnt, b, 1, "symmetric")
        ref_blk = x.copy(); orc.relax_block_gauss_seidel(op, ref_blk, b, Dinv, bs, 1, "symmetric")
        dM = DeviceMatrix(op)
        db = capi.DeviceArray.from_host(b)
        dD = capi.DeviceArray.from_host(Dinv.reshape(-1))
        for kw in (dict(gs_mode=0, flow_cap=0), dict(flow_cap=32), dict(flow_cap=256), dict(gs_mode=1), dict(gs_mode=2, gran_cap=0),
                   dict(gs_mode=2, gran_cap=3), dict(gs_mode=4, gran_cap=0), dict(gs_mode=4, gran_cap=5), dict(gs_mode=3),
                   dict(gs_mode=0, flow_cap=32, lds_entries=384), dict(gs_mode=2, lds_entries=96)):
            dM.tune(**kw)
            dx = capi.DeviceArray.from_host(x)
            dM.gauss_seidel(dx, db, sweep="symmetric")
            assert np.array_equal(dx.download(), ref_pnt), (kw, bs)
            dx.upload(x)
            dM.block_gauss_seidel(dx, db, dD, sweep="symmetric")
            assert np.array_equal(dx.download(), ref_blk), (kw, bs)
            assert not dM.flow_error(), (kw, bs)
        dM.tune(lds_entries=1536)


def test_block_gauss_seidel_fast_order_agrees_to_rounding():
    """The lane-parallel fast order of the block Gauss-Seidel sweep (tune gs_order = 1; csrc/pamg_blane.hip: one persistent launch, lanes share a
    block row, component-wise sentinel hand-off) against the order-exact device sweep (= the reference's bits: amg_core::block_gauss_seidel,
    relaxation.h:1242-1298): 1e-13 per call on 2x2 blocks on a 3-D grid (several block rows per wave, the chip-wide static form), a structurally
    NON-symmetric 3x3 pattern with missing diagonal blocks (old values from a snapshot), 27-point 3x3 and 6x6 blocks (one-XCD ticket form),
    4x4 blocks with block rows of more than 64 blocks (two blocks per lane); forward, backward, symmetric, two iterations; the same bits on a
    second run; 5x5 blocks (not compiled) stay with the exact kernels; the BSR point sweep: exact order bit for bit, fast order through the scalar twin."""
    from oracle import oracle as orc
    from tools.problems import poisson_csr
    rng = np.random.RandomState(31)
    def bsr_of(P, bs, diag_boost, drop=False):
        P = sp.csr_array(P); P.sort_indices()
        nb = P.shape[0]
        dat = (rng.rand(P.nnz, bs, bs) - 0.5) * 0.3
        rows = np.repeat(np.arange(nb), np.diff(P.indptr))
        dat[rows == P.indices] += diag_boost * np.eye(bs)
        return sp.bsr_array((dat, P.indices.astype(np.int32), P.indptr.astype(np.int32)), shape=(bs * nb, bs * nb), blocksize=(bs, bs))
    P7 = poisson_csr((60, 50, 40))
    g27 = poisson_csr((14, 12, 10)); g27 = sp.csr_array(((g27 @ g27 @ g27) != 0).astype(float))          # a wide 3-D stencil: ~60 blocks per block row
    Pn = sp.random(900, 900, density=0.01, random_state=rng, format="lil")
    for i in range(900):
        if i % 11 != 5:
            Pn[i, i] = 1.0
    long_rows = sp.csr_array(sp.random(400, 400, density=0.22, random_state=rng, format="csr") + sp.eye_array(400))
    cases = [("2x2 grid", bsr_of(P7, 2, 6.0)), ("3x3 nonsym", bsr_of(Pn, 3, 5.0)), ("3x3 wide", bsr_of(g27, 3, 12.0)), ("6x6 wide", bsr_of(g27, 6, 20.0)),
             ("4x4 long", bsr_of(long_rows, 4, 40.0)), ("5x5", bsr_of(poisson_csr((9, 8, 7)), 5, 6.0))]
    for name, M in cases:
        bs = M.blocksize[0]
        op = sparse_op(M)
        n = op.shape[0]; nb = n // bs
        x, b = rng.rand(n), rng.rand(n)
        Dinv = np.zeros((nb, bs, bs))
        rows = np.repeat(np.arange(nb), np.diff(M.indptr))
        dm = rows == M.indices
        Dinv[rows[dm]] = np.linalg.inv(M.data[dm])
        dM = DeviceMatrix(op)
        db = capi.DeviceArray.from_host(b)
        dD = capi.DeviceArray.from_host(Dinv.reshape(-1))
        ref_pnt = x.copy(); orc.relax_gauss_seidel(op, ref_pnt, b, 1, "symmetric")
        for sweep in ("forward", "backward", "symmetric"):
            ref = x.copy(); orc.relax_block_gauss_seidel(op, ref, b, Dinv, bs, 2, sweep)
            out = {}
            for order in (0, 1, 1):
                dM.tune(gs_order=order)
                dx = capi.DeviceArray.from_host(x)
                dM.block_gauss_seidel(dx, db, dD, sweep=sweep, iterations=2)
                got = dx.download()
                assert not dM.flow_error(), (name, sweep, order)
                if order == 0:
                    assert np.array_equal(got, ref), (name, sweep)
                else:
                    which = 1 if sweep == "backward" else 0
                    if bs != 5:
                        assert dM.lane_info(which)["groups"] > 0, (name, sweep)                # the lane form really ran
                    else:
                        assert dM.lane_info(which)["groups"] == 0 and np.array_equal(got, ref)
                    if 1 in out:
                        assert np.array_equal(out[1], got), (name, sweep)                       # reproducible
                out[order] = got
            err = np.max(np.abs(out[0] - out[1]))
            assert err <= 1e-13 * max(1.0, np.max(np.abs(out[0]))), (name, sweep, err)
        # the BSR POINT sweep (amg_core::bsr_gauss_seidel): exact order = the reference's bits; fast order (round 6) = the scalar sweep of the
        # flattened rows on the operator's CSR twin, lane-parallel: 1e-13, reproducible
        dM.tune(gs_order=0)
        dx = capi.DeviceArray.from_host(x)
        dM.gauss_seidel(dx, db, sweep="symmetric")
        assert np.array_equal(dx.download(), ref_pnt), name
        assert dM.point_twin() == 0
        dM.tune(gs_order=1)
        got = []
        for _ in range(2):
            dx = capi.DeviceArray.from_host(x)
            dM.gauss_seidel(dx, db, sweep="symmetric")
            got.append(dx.download())
            assert not dM.flow_error(), name
        assert dM.point_twin() in (1, 2), name
        if name in ("2x2 grid", "3x3 wide"):                                                     # (6x6 wide: 360 entries per flattened row, beyond the lane forms)
            assert dM.point_twin() == 1, name                                                        # a fast-order form really ran
        assert np.array_equal(got[0], got[1]), name
        assert np.max(np.abs(got[0] - ref_pnt)) <= 1e-13 * max(1.0, np.max(np.abs(ref_pnt))), name
        for sweep in ("forward", "backward"):
            ref1 = x.copy(); orc.relax_gauss_seidel(op, ref1, b, 2, sweep)
            dx = capi.DeviceArray.from_host(x)
            dM.gauss_seidel(dx, db, sweep=sweep, iterations=2)
            assert np.max(np.abs(dx.download() - ref1)) <= 1e-13 * max(1.0, np.max(np.abs(ref1))), (name, sweep)
        dM.free()


def test_indexed_jacobi_bit_exact():
    """jacobi_indexed (Layer 1, the amg_core twin) and the cf_jacobi / fc_jacobi wrappers (resident row-subset
    operators, csr_stream_kernel<JACOBI_IDX> + scatter) vs the reference's outputs in kernels_indexed.npz --
    bit for bit, f64 and f32, incl. an empty row and a missing diagonal; plus the wrappers' error contract."""
    from conftest import GOLDEN
    import pyamg_amd.amg_core as gcore
    z = np.load(GOLDEN / "kernels_indexed.npz")
    keys = sorted({k.split(".")[0] for k in z.files})
    for k in keys:
        Ap, Aj, Ax = z[f"{k}.indptr"].astype(np.int32), z[f"{k}.indices"].astype(np.int32), z[f"{k}.data"]
        n = Ap.size - 1
        M = sp.csr_array((Ax, Aj, Ap), shape=(n, n))
        x, b, idx, F, Cp = (z[f"{k}.{t}"] for t in ("x", "b", "idx", "F", "C"))
        y = x.copy(); gcore.jacobi_indexed(Ap, Aj, Ax, y, b, idx, np.array([0.7], dtype=Ax.dtype))
        assert np.array_equal(y, z[f"{k}.jacobi_indexed"]), k
        y = x.copy(); grelax.jacobi_indexed(M, y, b, idx, iterations=1, omega=0.7)
        assert np.array_equal(y, z[f"{k}.jacobi_indexed"]), k
        y = x.copy(); grelax.fc_jacobi(M, y, b, Cp, F, iterations=2, f_iterations=2, c_iterations=1, omega=0.9)
        assert np.array_equal(y, z[f"{k}.fc_jacobi"]), k
        y = x.copy(); grelax.cf_jacobi(M, y, b, Cp, F, iterations=1, f_iterations=1, c_iterations=2, omega=1.0)
        assert np.array_equal(y, z[f"{k}.cf_jacobi"]), k
    with pytest.raises(ValueError):
        grelax.jacobi_indexed(M, x.copy(), b, np.array([0, n], dtype=np.int32))          # relaxation.py:1108-1109
    with pytest.raises(TypeError):
        gcore.jacobi_indexed(Ap, Aj, Ax, x.copy(), b, idx.astype(np.int64), np.array([0.7], dtype=Ax.dtype))


def test_schwarz_bit_exact():
    """overlapping_schwarz_csr (Layer 1 twin) and relaxation.schwarz: subdomains of one dependency level side by side, one
    wave each, the reference's arithmetic order -- vs the reference's outputs in kernels_schwarz.npz, bit for bit: all
    sweep directions, a strided sweep, large two-hop subdomains, f64 and f32."""
    from conftest import GOLDEN
    import pyamg_amd.amg_core as gcore
    z = np.load(GOLDEN / "kernels_schwarz.npz")
    keys = sorted({k.split(".")[0] for k in z.files})
    assert len(keys) == 5
    for k in keys:
        Ap, Aj, Ax = z[f"{k}.indptr"].astype(np.int32), z[f"{k}.indices"].astype(np.int32), z[f"{k}.data"]
        n = Ap.size - 1
        M = sp.csr_array((Ax, Aj, Ap), shape=(n, n))
        x, b, sub, sptr, inv, iptr = (z[f"{k}.{t}"] for t in ("x", "b", "sub", "sptr", "inv", "iptr"))
        if k.endswith("_big"):
            y = x.copy(); grelax.schwarz(M, y, b, iterations=1, subdomain=sub, subdomain_ptr=sptr, sweep="symmetric")
            assert np.array_equal(y, z[f"{k}.symmetric"]), k
            continue
        for sweep in ("forward", "backward", "symmetric"):
            y = x.copy(); grelax.schwarz(sp.csr_array(M.copy()), y, b, iterations=2, sweep=sweep)
            assert np.array_equal(y, z[f"{k}.{sweep}"]), (k, sweep)
        y = x.copy()
        gcore.overlapping_schwarz_csr(Ap, Aj, Ax, y, b, inv, iptr.astype(np.int32), sub.astype(np.int32), sptr.astype(np.int32),
                                      len(sptr) - 1, n, 3, n - 1, 2)
        assert np.array_equal(y, z[f"{k}.strided"]), k
    with pytest.raises(ValueError):
        grelax.schwarz(M, x.copy(), b, inv_subblock=inv)
    with pytest.raises(ValueError):
        grelax.schwarz(M, x.copy(), b, sweep="sideways")


def test_schwarz_persistent_sweep_is_the_level_launches_and_the_oracle():
    """the ONE persistent launch per Schwarz sweep (waves walk the subdomains in level order and wait on the counter of the level before)
    against one launch per dependency level and against the oracle's sequential sweep (relaxation.h:1420-1492): the same bits, on operators with
    hundreds of dependency levels (2-D Poisson: 4 n levels), in both directions, strided, f64 and f32; the error word stays clear."""
    import ctypes as C
    from oracle import oracle as orc
    from pyamg_amd import _capi as capi
    from pyamg_amd.hierarchy import sparse_op
    from pyamg_amd.multilevel import DeviceMatrix
    from tools.problems import poisson_csr
    lib = capi.lib()
    rng = np.random.RandomState(11)
    for grid, dtype in (((90, 90), np.float64), ((20, 20, 20), np.float64), ((70, 70), np.float32)):
        A = sp.csr_matrix(poisson_csr(grid)).astype(dtype)
        A.sort_indices()
        n = A.shape[0]
        sub, sptr, inv, iptr = grelax.schwarz_parameters(A)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)       # noqa: E731
        Sp, Sj, Tp, Tx = i32(sptr), i32(sub), i32(iptr), np.ascontiguousarray(inv, dtype=dtype)
        dA = DeviceMatrix(sparse_op(A))
        h = C.c_void_p()
        capi.check(lib.pamg_schwarz_create(C.byref(h), dA.handle, n, capi.ptr(Sp), capi.ptr(Sj), capi.ptr(Tp), capi.ptr(Tx)), "pamg_schwarz_create")
        x0, b = rng.rand(n).astype(dtype), rng.rand(n).astype(dtype)
        bd = capi.DeviceArray.from_host(b)
        for (r0, r1, rs) in ((0, n, 1), (n - 1, -1, -1), (3, n - 1, 2)):
            want = x0.copy()
            orc.overlapping_schwarz_csr(i32(A.indptr), i32(A.indices), A.data, want, b, Tx, Tp, Sj, Sp, r0, r1, rs)
            got = []
            for mode in (0, 1):
                capi.check(lib.pamg_schwarz_set_mode(h, mode), "pamg_schwarz_set_mode")
                xd = capi.DeviceArray.from_host(x0)
                for _ in range(2):                                    # a second sweep on the result: counters are reset per sweep
                    capi.check(lib.pamg_schwarz_sweep(h, xd.ptr, bd.ptr, r0, r1, rs, None), "pamg_schwarz_sweep")
                err = C.c_int(0)
                capi.check(lib.pamg_schwarz_error(h, C.byref(err)), "pamg_schwarz_error")
                assert err.value == 0
                got.append(xd.download())
            want2 = want.copy()
            orc.overlapping_schwarz_csr(i32(A.indptr), i32(A.indices), A.data, want2, b, Tx, Tp, Sj, Sp, r0, r1, rs)
            assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], want2), (grid, dtype, r0, rs)
        info = (C.c_int64 * 4)()
        capi.check(lib.pamg_schwarz_info(h, info), "pamg_schwarz_info")
        assert info[2] >= 2 * grid[0] and info[3] >= 2 * grid[0]      # long chains: what the persistent form is for
        capi.check(lib.pamg_schwarz_destroy(h), "pamg_schwarz_destroy")
        dA.free()


def test_indexed_gauss_seidel_bit_exact():
    """gauss_seidel_indexed (Layer 1 twin and the relaxation wrapper): the listed rows in list order as a forward sweep of
    the renumbered operator -- vs the reference's outputs in kernels_gsidx.npz, bit for bit: arbitrary order, rows listed
    two and three times, a strided slice of the list, forward / backward / symmetric, f64 and f32."""
    from conftest import GOLDEN
    import pyamg_amd.amg_core as gcore
    z = np.load(GOLDEN / "kernels_gsidx.npz")
    keys = sorted({k.split(".")[0] for k in z.files})
    assert len(keys) == 4
    for k in keys:
        Ap, Aj, Ax = z[f"{k}.indptr"].astype(np.int32), z[f"{k}.indices"].astype(np.int32), z[f"{k}.data"]
        n = Ap.size - 1
        M = sp.csr_array((Ax, Aj, Ap), shape=(n, n))
        x, b, idx, dup = (z[f"{k}.{t}"] for t in ("x", "b", "idx", "dup"))
        for sweep in ("forward", "backward", "symmetric"):
            y = x.copy(); grelax.gauss_seidel_indexed(M, y, b, idx, iterations=2, sweep=sweep)
            assert np.array_equal(y, z[f"{k}.{sweep}"]), (k, sweep)
        y = x.copy(); grelax.gauss_seidel_indexed(M, y, b, dup, iterations=1, sweep="forward")
        assert np.array_equal(y, z[f"{k}.dup.forward"]), k
        y = x.copy(); gcore.gauss_seidel_indexed(Ap, Aj, Ax, y, b, idx, 1, len(idx) - 1, 2)
        assert np.array_equal(y, z[f"{k}.strided"]), k
    with pytest.raises(ValueError):
        grelax.gauss_seidel_indexed(M, x.copy(), b, np.array([0, n], dtype=np.int32))
    with pytest.raises(ValueError):
        grelax.gauss_seidel_indexed(M, x.copy(), b, idx, sweep="sideways")
    with pytest.raises(TypeError):
        gcore.gauss_seidel_indexed(Ap, Aj, Ax, x.copy(), b, idx.astype(np.int64), 0, 3, 1)


def test_indexed_block_jacobi_bit_exact():
    """block_jacobi_indexed (Layer 1, the amg_core twin) and the cf_block_jacobi / fc_block_jacobi wrappers (one full
    block-Jacobi step + take-over of the listed block rows) vs the reference's outputs in kernels_blockidx.npz -- bit for
    bit, 3x3 and 2x2 blocks, f64 and f32, incl. an empty block row; plus the error contract."""
    from conftest import GOLDEN
    import pyamg_amd.amg_core as gcore
    z = np.load(GOLDEN / "kernels_blockidx.npz")
    keys = sorted({k.split(".")[0] for k in z.files})
    assert len(keys) == 4
    for k in keys:
        Ap, Aj, Ax = z[f"{k}.indptr"].astype(np.int32), z[f"{k}.indices"].astype(np.int32), z[f"{k}.data"]
        bs = Ax.shape[1]
        nb = Ap.size - 1
        M = sp.bsr_array((Ax, Aj, Ap), shape=(nb * bs, nb * bs))
        x, b, idx, F, Cp, Dinv = (z[f"{k}.{t}"] for t in ("x", "b", "idx", "F", "C", "Dinv"))
        y = x.copy()
        gcore.block_jacobi_indexed(Ap, Aj, np.ravel(Ax), y, b, np.ravel(Dinv), idx, np.array([0.7], dtype=Ax.dtype), bs)
        assert np.array_equal(y, z[f"{k}.block_jacobi_indexed"]), k
        y = x.copy(); grelax.fc_block_jacobi(M, y, b, Cp, F, Dinv=Dinv, blocksize=bs, iterations=2, f_iterations=2, c_iterations=1, omega=0.9)
        assert np.array_equal(y, z[f"{k}.fc_block_jacobi"]), k
        y = x.copy(); grelax.cf_block_jacobi(M, y, b, Cp, F, Dinv=Dinv, blocksize=bs, iterations=1, f_iterations=1, c_iterations=2, omega=1.0)
        assert np.array_equal(y, z[f"{k}.cf_block_jacobi"]), k
    with pytest.raises(ValueError):
        grelax.cf_block_jacobi(M, x.copy(), b, np.array([nb], dtype=np.int32), F, Dinv=Dinv, blocksize=bs)
    # blocksize 1 = the point method on the scalar rows (what the reference's smoother setup substitutes, smoothing.py:731-734)
    y1, y2 = x.copy(), x.copy()
    Cs, Fs = np.arange(0, nb * bs, 2, dtype=np.int32), np.arange(1, nb * bs, 2, dtype=np.int32)
    grelax.cf_block_jacobi(M, y1, b, Cs, Fs, blocksize=1, iterations=2, omega=0.8)
    grelax.cf_jacobi(M.tocsr(), y2, b, Cs, Fs, iterations=2, omega=0.8)
    assert np.array_equal(y1, y2) and not np.array_equal(y1, x)
    y1, y2 = x.copy(), x.copy()
    grelax.block_gauss_seidel(M, y1, b, blocksize=1, sweep="symmetric")
    grelax.gauss_seidel(M.tocsr(), y2, b, sweep="symmetric")
    assert np.array_equal(y1, y2)
    y1, y2 = x.copy(), x.copy()
    grelax.block_jacobi(M, y1, b, blocksize=1, omega=0.7, iterations=2)
    grelax.jacobi(M.tocsr(), y2, b, omega=0.7, iterations=2)
    assert np.array_equal(y1, y2)
    with pytest.raises(ValueError):
        grelax.block_jacobi(M, x.copy(), b, Dinv=Dinv, blocksize=1)
    with pytest.raises(TypeError):
        gcore.block_jacobi_indexed(Ap, Aj, np.ravel(Ax), x.copy(), b, np.ravel(Dinv), idx.astype(np.int64), np.array([0.7], dtype=Ax.dtype), bs)


def test_normal_equation_smoothers_bit_exact():
    """gauss_seidel_ne (Kaczmarz), gauss_seidel_nr and jacobi_ne on the device (order-exact level schedules
    over shared indices; (omega A)^T SpMV) vs the reference's outputs in kernels_ne.npz -- bit for bit, all
    sweep directions, incl. an empty row and an empty column."""
    from conftest import GOLDEN
    z = np.load(GOLDEN / "kernels_ne.npz")
    for tag in ("irr", "pois"):
        n = z[f"{tag}.indptr"].size - 1
        M = sp.csr_array((z[f"{tag}.data"], z[f"{tag}.indices"].astype(np.int32), z[f"{tag}.indptr"].astype(np.int32)), shape=(n, n))
        x, b = z[f"{tag}.x"], z[f"{tag}.b"]
        for sweep in ("forward", "backward", "symmetric"):
            y = x.copy(); grelax.gauss_seidel_ne(M, y, b, iterations=2, sweep=sweep, omega=0.9)
            assert np.array_equal(y, z[f"{tag}.gauss_seidel_ne.{sweep}"]), (tag, sweep)
            y = x.copy(); grelax.gauss_seidel_nr(sp.csc_array(M), y, b, iterations=2, sweep=sweep, omega=1.1)
            assert np.array_equal(y, z[f"{tag}.gauss_seidel_nr.{sweep}"]), (tag, sweep)
        y = x.copy(); grelax.jacobi_ne(M, y, b, iterations=3, omega=0.6)
        assert np.array_equal(y, z[f"{tag}.jacobi_ne"]), tag
    with pytest.raises(ValueError):
        grelax.gauss_seidel_nr(M, x.copy(), b, sweep="sideways")
    # both schedulers of the Kaczmarz sweeps: one persistent workgroup (default on these narrow schedules) and one
    # launch per dependency level (gs_mode = 1)
    from pyamg_amd.hierarchy import _normal_equation_spec as _nes
    ne0 = _nes("gauss_seidel_ne", M, 1, "forward", 0.9)
    dA = DeviceMatrix(sparse_op(M))
    dD = capi.DeviceArray.from_host(ne0.Dinv)
    db = capi.DeviceArray.from_host(b)
    for mode in (0, 1):
        dA.tune(gs_mode=mode)
        dx = capi.DeviceArray.from_host(x)
        dA.kaczmarz(dx, dD, 0.9, "symmetric", 2, b=db)
        assert np.array_equal(dx.download(), z[f"{tag}.gauss_seidel_ne.symmetric"]), mode
    # Layer 1: the amg_core twins on host buffers, driven exactly as the reference's wrappers drive amg_core
    import pyamg_amd.amg_core as gcore
    from pyamg_amd.hierarchy import _normal_equation_spec
    ne = _normal_equation_spec("gauss_seidel_ne", M, 1, "forward", 0.9)
    nr = _normal_equation_spec("gauss_seidel_nr", M, 1, "forward", 1.1)
    y = x.copy()
    for _ in range(2):
        gcore.gauss_seidel_ne(M.indptr, M.indices, M.data, y, b, 0, n, 1, ne.Dinv, 0.9)
    assert np.array_equal(y, z[f"{tag}.gauss_seidel_ne.forward"])
    y = x.copy(); r = b - M @ y
    for _ in range(2):
        gcore.gauss_seidel_nr(nr.At.indptr, nr.At.indices, nr.At.data, y, r, n - 1, -1, -1, nr.Dinv, 1.1)
    assert np.array_equal(y, z[f"{tag}.gauss_seidel_nr.backward"])
    y = x.copy(); temp = np.zeros(n)
    jn = _normal_equation_spec("jacobi_ne", M, 1, "forward", 0.6)
    for _ in range(3):
        delta = (np.ravel(b - M @ y) * jn.Dinv).astype(M.dtype)
        gcore.jacobi_ne(M.indptr, M.indices, M.data, y, b, delta, temp, 0, n, 1, np.array([0.6]))
    assert np.array_equal(y, z[f"{tag}.jacobi_ne"])


def test_kaczmarz_fast_order_agrees_to_rounding():
    """The lane-parallel fast order of the Kaczmarz sweeps (tune gs_order = 1 on the operator handed to pamg_matrix_kaczmarz; csrc/pamg_kz.hip:
    one persistent launch, lanes share a line, versioned 16-byte slots hand the rewritten vector over) against the order-exact device sweeps
    (= the reference's bits: amg_core::gauss_seidel_ne / gauss_seidel_nr, relaxation.h:875-904, 939-975): 1e-13 per call on upwind
    convection-diffusion (short lines) and a dense-ish operator (long lines, two slots per lane); forward, backward, symmetric, two iterations; the
    same bits on a second run; the exact kernels untouched by the switch.  Both device orders are also held against the ORACLE's restatement of the
    reference loops (oracle.gauss_seidel_ne / gauss_seidel_nr run in the same sequence on the same running arrays): exact order bit for bit, fast
    order 1e-13 (the chain device-exact == reference was only pinned on other matrices)."""
    from oracle import oracle as orc
    from pyamg_amd.hierarchy import _normal_equation_spec as _nes
    rng = np.random.RandomState(29)
    n = 4000
    conv = sp.csr_array(sp.diags_array([np.full(n, 4.0), -np.ones(n - 1), -2 * np.ones(n - 60), -np.ones(n - 60)], offsets=[0, -1, -60, 60], shape=(n, n)))
    dense = sp.csr_array(sp.random(900, 900, density=0.1, random_state=rng, format="csr") + sp.diags_array(np.full(900, 9.0)))
    for M in (conv, dense):
        M.sort_indices()
        m = M.shape[0]
        x, b = rng.rand(m), rng.rand(m)
        for kind, omega in (("gauss_seidel_ne", 0.9), ("gauss_seidel_nr", 1.1)):
            spec = _nes(kind, M, 1, "forward", omega)
            Lop = sparse_op(M) if kind == "gauss_seidel_ne" else spec.At
            dL = DeviceMatrix(Lop)
            dD = capi.DeviceArray.from_host(spec.Dinv)
            db = capi.DeviceArray.from_host(b)
            for sweep in ("forward", "backward", "symmetric"):
                out = {}
                for order in (0, 1, 1):
                    dL.tune(gs_order=order)
                    if kind == "gauss_seidel_ne":
                        dv = capi.DeviceArray.from_host(x)
                        dL.kaczmarz(dv, dD, omega, sweep, 2, b=db)
                        got = (dv.download(),)
                    else:
                        dr = capi.DeviceArray.from_host(b - M @ x)
                        dxo = capi.DeviceArray.from_host(x)
                        dL.kaczmarz(dr, dD, omega, sweep, 2, xout=dxo)
                        got = (dxo.download(), dr.download())
                    assert not dL.flow_error()
                    if order == 1:
                        assert dL.kz_info(0)["groups"] > 0, (kind, sweep)                    # the lane form really ran
                        if 1 in out:
                            assert all(np.array_equal(a_, b_) for a_, b_ in zip(out[1], got))      # reproducible
                    out[order] = got
                for a_, b_ in zip(out[0], out[1]):
                    assert np.max(np.abs(a_ - b_)) <= 1e-13 * max(1.0, np.max(np.abs(a_))), (kind, sweep, np.max(np.abs(a_ - b_)))
                # the oracle's loops in the same sequence: `2` iterations of the sweep on the running vectors
                dirs = {"forward": [(0, m, 1)], "backward": [(m - 1, -1, -1)], "symmetric": [(0, m, 1), (m - 1, -1, -1)]}[sweep] * 2
                if kind == "gauss_seidel_ne":
                    xo = x.copy()
                    for (r0, r1, rs) in dirs:
                        orc.gauss_seidel_ne(Lop.indptr, Lop.indices, Lop.data, xo, b, r0, r1, rs, spec.Dinv, omega)
                    want = (xo,)
                else:
                    xo, ro = x.copy(), b - M @ x
                    for (r0, r1, rs) in dirs:
                        orc.gauss_seidel_nr(Lop.indptr, Lop.indices, Lop.data, xo, ro, r0, r1, rs, spec.Dinv, omega)
                    want = (xo, ro)
                for w_, e_, f_ in zip(want, out[0], out[1]):
                    assert np.array_equal(w_, e_), (kind, sweep)                                   # exact order: the oracle's bits
                    assert np.max(np.abs(w_ - f_)) <= 1e-13 * max(1.0, np.max(np.abs(w_))), (kind, sweep, np.max(np.abs(w_ - f_)))
            dL.free()


def test_layer1_operator_cache():
    """Layer 1 keeps the last operators resident: the same three arrays again -> no new entry; the same arrays with
    CHANGED contents -> the stale copy is replaced and the sweep uses the new values; other arrays -> another entry.
    Results equal the oracle's in every case."""
    import ctypes as C
    from oracle import oracle as orc
    from pyamg_amd import _capi as capi
    from tools.problems import poisson_csr
    lib = capi.lib()

    def size():
        k = C.c_int(0)
        capi.check(lib.pamg_l1_cache_size(C.byref(k)), "pamg_l1_cache_size")
        return k.value

    capi.check(lib.pamg_l1_cache_clear(), "pamg_l1_cache_clear")
    A = poisson_csr((30, 30))
    n = A.shape[0]
    Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    rng = np.random.RandomState(3)
    x0, b = rng.rand(n), rng.rand(n)
    for trial in range(3):
        x, ref = x0.copy(), x0.copy()
        gcore.gauss_seidel(Ap, Aj, Ax, x, b, 0, n, 1)
        orc.gauss_seidel(Ap, Aj, Ax, ref, b, 0, n, 1)
        assert np.array_equal(x, ref)
        assert size() == 1
        if trial == 1:
            Ax[::3] *= 1.5                                  # in place: same addresses, new contents
    B = poisson_csr((17, 19))
    Bp, Bj, Bx = B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.copy()
    m = B.shape[0]
    y, yref = rng.rand(m), None
    yref = y.copy()
    c = rng.rand(m)
    gcore.gauss_seidel(Bp, Bj, Bx, y, c, m - 1, -1, -1)
    orc.gauss_seidel(Bp, Bj, Bx, yref, c, m - 1, -1, -1)
    assert np.array_equal(y, yref) and size() == 2
    capi.check(lib.pamg_l1_cache_clear(), "pamg_l1_cache_clear")
    assert size() == 0


def test_pybind11_module_runs_the_same_kernels(kernels_npz):
    """the pybind11 module and the ctypes twin are two faces of the same C ABI: identical bits, equal to the oracle"""
    from pyamg_amd import _build
    if not _build.pybind_path().exists():
        pytest.skip("pybind11 module not built")
    from oracle import oracle as orc
    from pyamg_amd import _amg_core_pybind as pb
    from tools.problems import poisson_csr
    A = poisson_csr((21, 13))
    n = A.shape[0]
    Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    rng = np.random.RandomState(8)
    x0, b = rng.rand(n), rng.rand(n)
    for (r0, r1, rs) in ((0, n, 1), (n - 1, -1, -1)):
        xa, xb, xr = x0.copy(), x0.copy(), x0.copy()
        pb.gauss_seidel(Ap, Aj, Ax, xa, b, r0, r1, rs)
        gcore.gauss_seidel(Ap, Aj, Ax, xb, b, r0, r1, rs)
        orc.gauss_seidel(Ap, Aj, Ax, xr, b, r0, r1, rs)
        assert np.array_equal(xa, xr) and np.array_equal(xb, xr)
    y = np.zeros(n)
    pb.csr_matvec(n, n, Ap, Aj, Ax, x0, y)
    assert np.array_equal(y, A @ x0)
    om = np.array([0.7])
    xa, xr, t1, t2 = x0.copy(), x0.copy(), np.zeros(n), np.zeros(n)
    pb.jacobi(Ap, Aj, Ax, xa, b, t1, 0, n, 1, om)
    orc.jacobi(Ap, Aj, Ax, xr, b, t2, 0, n, 1, om[0])
    assert np.array_equal(xa, xr)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_8bit_value_codes_are_bit_identical(dtype):
    """Operators with at most 256 distinct values (the gallery's stencils: 2) stream one byte per value and look the
    value up in an LDS dictionary (tune key 21 switches back to the values themselves): same bits either way -- on the
    7-point stencil, with exactly 256 and with 257 distinct values (no codes), with +0 / -0 / explicit zeros as
    separate dictionary entries, with one row longer than the LDS window, and on a BSR(1,1) operator.  Operators with
    codes run as the row-gather kernel (lane = row, tune key 22) -- checked against the staged kernel on the codes and on the
    values as stored, every whole-operator epilogue."""
    import scipy.sparse as sp
    from tools.problems import poisson_csr
    rng = np.random.RandomState(8)
    P3 = poisson_csr((48, 48, 48))
    nn = 30000

    def banded(values):
        cols = (np.arange(nn)[:, None] + np.array([-40, -1, 0, 1, 40])[None, :]) % nn
        data = rng.choice(np.asarray(values, dtype=np.float64), size=cols.size)
        A = sp.csr_array((data, cols.ravel().astype(np.int32), np.arange(0, cols.size + 1, 5, dtype=np.int32)), shape=(nn, nn))
        return A
    v256 = np.concatenate([[0.0, -0.0], rng.randn(254)])
    A256, A257 = banded(v256), banded(np.concatenate([v256, [7.25]]))
    A257.data[:257] = np.concatenate([v256, [7.25]])            # all 257 present for sure
    A256.data[:256] = v256
    long_row = banded([1.0, -2.0, 0.5]).tolil()
    long_row[17, :6000] = rng.choice([3.0, -1.5], size=6000)
    long_row = long_row.tocsr()
    # a stencil with 400 rows of their own (two entries of each changed to one of 20 x 20 value pairs): more lists than the
    # table holds -> the rows outside it take the code arrays inside the row-pattern kernel
    odd = P3.tolil()
    extra = 1.0 + np.arange(20) / 32.0
    for k in range(400):
        i = 5000 + 37 * k
        odd[i, i - 1] = -extra[k % 20]
        odd[i, i + 1] = -extra[k // 20]
    odd = sp.csr_array(odd.tocsr())
    cases = [(P3, 2, 27), (A256, 256, 0), (A257, 0, 0), (long_row, 5, None), (sp.bsr_array(poisson_csr((300, 300)), blocksize=(1, 1)), 2, 9),
             (odd, 21, 245 if dtype == np.float64 else 255)]       # 8-entry lists: 245 of them fill the 24 KB table in f64
    for A, expect, npat in cases:
        A = A.astype(dtype)
        n = A.shape[0]
        x, b = rng.rand(n).astype(dtype), rng.rand(n).astype(dtype)
        dA = DeviceMatrix(sparse_op(A))
        if npat is None:                            # 3 values on 5 diagonals: up to 243 lists + the long row, which no list holds
            npat = dA.row_patterns()
            assert npat > 100
        assert dA.value_codes() == expect and dA.row_patterns() == npat, (dA.value_codes(), dA.row_patterns())
        dx, db = capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(b)
        out = {}
        # 5: row-pattern table kernel, one row per lane, 3: the default where a table exists (row masks where the lists allow it,
        # else the table kernel), 2: codes + row-gather kernel, 1: codes + staged kernel, 0: values as stored
        for flag in (5, 3, 2, 1, 0):
            dA.tune(val8=min(flag, 1), rowgather=int(flag >= 2), rowpat={5: 3, 3: 1}.get(flag, 0))
            assert dA.value_codes() == (expect if flag else 0) and dA.row_patterns() == (npat if flag >= 3 else 0)
            dy = capi.DeviceArray(n, dtype)
            dA.spmv(capi.SPMV_RESID, dx, dy, b=db)
            dz = capi.DeviceArray(n, dtype)
            dA.spmv(capi.SPMV_SET, dx, dz)
            dj = capi.DeviceArray.from_host(x)
            dw = capi.DeviceArray(n, dtype)
            dA.jacobi(dj, db, dw, 0.8, iterations=2)
            out[flag] = (dy.download(), dz.download(), dj.download())
        for k in range(3):
            assert all(np.array_equal(out[0][k], out[f][k], equal_nan=True) for f in (1, 2, 3, 5)), k
        # the remaining epilogues of the row-gather kernel against the staged kernel on the values as stored
        dA.tune(val8=1, rowgather=1, rowpat=1)
        res = {}
        for flag in (4, 2, 1, 0):
            dA.tune(val8=min(flag, 1), rowgather=min(flag, 1), rowpat={4: 3, 2: 1}.get(flag, 0))
            dy = capi.DeviceArray.from_host(b)
            dA.spmv(capi.SPMV_ACC, dx, dy)
            d2 = capi.DeviceArray.from_host(x)
            dA.spmv(capi.SPMV_ACC_AXPBY, dx, d2, b=db, c=-1.7)
            d3 = capi.DeviceArray(n, dtype)
            dA.spmv(capi.SPMV_AXPBY, dx, d3, b=db, c=0.37)
            o = capi.DeviceArray(1, np.float64)
            dA.resid_sumsq(dx, db, o)
            res[flag] = (dy.download(), d2.download(), d3.download(), o.download())
        for k in range(4):
            assert np.array_equal(res[0][k], res[1][k], equal_nan=True) and np.array_equal(res[0][k], res[2][k], equal_nan=True), k
            assert np.array_equal(res[0][k], res[4][k], equal_nan=True), k
        dA.tune(val8=1, rowgather=1, rowpat=1)
        if dtype == np.float64:
            assert np.array_equal(out[1][1], sp.csr_array(A) @ x)
        dA.free()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_row_mask_kernels_are_bit_identical(dtype):
    """Row masks (tune key 23 = 1 where every list is the stencil's interior list with entries left out): the linear kernel
    (one row per lane, no table) under its three workgroup orders, nontemporal or not, +-1 by DPP or gathered, and the lattice
    kernel (64 x 4 x kz tiles, kz = 2, 4, 8, both workgroup orders) against the row-gather form on the same operator -- every
    epilogue, bit for bit; rows that are no sub-list walk the CSR arrays inside the kernels."""
    import scipy.sparse as sp
    from tools.problems import poisson_csr
    rng = np.random.RandomState(12)
    lat = poisson_csr((16, 32, 64))                      # 64-row lines, 32 lines per plane (XCD slabs: 8 tiles of 4 lines), 16 planes
    wide = poisson_csr((8, 8, 256))                      # four 64-row tiles per line, 2 tiles of lines: no XCD slabs
    odd = lat.tolil()
    for k in range(60):
        i = 900 + 41 * k
        odd[i, i + 7] = -0.5
        odd[i + 3, i + 2] = -1.25
    odd = sp.csr_array(odd.tocsr())
    flat = poisson_csr((128, 128))                       # 5 entries: linear form only
    for A, lattice in ((lat, True), (wide, True), (odd, True), (flat, False)):
        A = sp.csr_array(A).astype(dtype)
        n = A.shape[0]
        x, b = rng.rand(n).astype(dtype), rng.rand(n).astype(dtype)
        dA = DeviceMatrix(sparse_op(A))
        assert dA.row_patterns() > 0
        dx, db = capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(b)

        def everything():
            outs = []
            for mode, kw in ((capi.SPMV_RESID, dict(b=db)), (capi.SPMV_SET, {}), (capi.SPMV_AXPBY, dict(b=db, c=0.37))):
                dy = capi.DeviceArray(n, dtype)
                dA.spmv(mode, dx, dy, **kw)
                outs.append(dy.download())
                dy.free()
            dy = capi.DeviceArray.from_host(b)
            dA.spmv(capi.SPMV_ACC, dx, dy)
            outs.append(dy.download())
            dy.free()
            dy = capi.DeviceArray.from_host(x)
            dA.spmv(capi.SPMV_ACC_AXPBY, dx, dy, b=db, c=-1.7)
            outs.append(dy.download())
            dy.free()
            dj, dw = capi.DeviceArray.from_host(x), capi.DeviceArray(n, dtype)
            dA.jacobi(dj, db, dw, 0.8, iterations=2)
            outs.append(dj.download())
            dj.free(); dw.free()
            o = capi.DeviceArray(1, np.float64)
            dA.resid_sumsq(dx, db, o)                 # one partial per row range in every form: the norm's bits agree too
            outs.append(o.download())
            o.free()
            return outs
        dA.tune(rowpat=0)
        ref = everything()
        if dtype == np.float64:
            assert np.array_equal(ref[1], A @ x)
        variants = [dict(rowpat=4, rowmask_flags=f) for f in (0, 1, 2, 3, 4, 5)]
        if lattice:
            variants += [dict(rowpat=1, rowmask_kz=kz, rowmask_flags=f) for kz in (2, 4, 8) for f in (0, 1, 2, 3)]
        for v in variants:
            dA.tune(**v)
            got = everything()
            for k, (r, g) in enumerate(zip(ref, got)):
                assert np.array_equal(r, g, equal_nan=True), (v, k, int(np.sum(r != g)))
        dA.free()


def test_row_masks_on_a_row_shard():
    """A slab of a 3-D stencil in local numbering [owned | halo] (what pyamg_amd.dist ships to a rank): the interior ranges are
    one window of whole planes and run in the row-mask kernels (lattice form where the planes divide), the boundary ranges
    keep the range-list kernels; interior + boundary together = the global product's rows, bit for bit, in every form."""
    from tools.problems import poisson_csr
    rng = np.random.RandomState(21)
    nz, ny, nx = 44, 32, 64
    G = sp.csr_array(poisson_csr((nz, ny, nx)))
    P = ny * nx
    xg, bg = rng.rand(G.shape[0]), rng.rand(G.shape[0])
    for z0, z1 in ((2, 36), (2, 42), (0, 34)):         # 32 / 38 / 33 interior planes: lattice form with 8 / 2 planes per lane, linear form
        lo, hi = z0 * P, z1 * P
        rows = G[lo:hi].tocsr()
        halo = np.unique(rows.indices[(rows.indices < lo) | (rows.indices >= hi)])
        remap = -np.ones(G.shape[0], dtype=np.int64)
        remap[lo:hi] = np.arange(hi - lo)
        remap[halo] = (hi - lo) + np.arange(halo.size)
        L = sp.csr_array((rows.data, remap[rows.indices].astype(np.int32), rows.indptr), shape=(hi - lo, hi - lo + halo.size))
        xl = np.concatenate([xg[lo:hi], xg[halo]])
        ref_set, ref_res = (G @ xg)[lo:hi], (bg - G @ xg)[lo:hi]
        dA = DeviceMatrix(sparse_op(L))
        # (rows of the plane under the upper halo are regular rows when that halo directly follows the owned rows: z0 = 0)
        assert dA.row_patterns() > 0 and dA.row_masks()["entries"] == 7 and dA.row_masks()["walked_rows"] == (2 * P if z0 else 0)
        dA.split_ranges(hi - lo)
        dx, db = capi.DeviceArray.from_host(xl), capi.DeviceArray.from_host(bg[lo:hi])
        for tune in (dict(rowpat=1), dict(rowpat=1, rowmask_kz=4, rowmask_flags=0), dict(rowpat=4), dict(rowpat=3), dict(rowpat=0)):
            dA.tune(**tune)
            for mode, kw, ref in ((capi.SPMV_SET, {}, ref_set), (capi.SPMV_RESID, dict(b=db), ref_res)):
                dy = capi.DeviceArray.from_host(np.full(hi - lo, np.nan))
                dA.spmv(mode, dx, dy, part=1, **kw)
                part1 = dy.download()
                dA.spmv(mode, dx, dy, part=2, **kw)
                both = dy.download()
                assert np.isnan(part1).sum() > 0 and not np.isnan(both).any()          # the interior alone leaves the boundary rows untouched
                assert np.array_equal(both, ref), (z0, z1, tune, mode)
                dA.spmv(mode, dx, dy, **kw)                                           # every range in one launch
                assert np.array_equal(dy.download(), ref)
                dy.free()
        dA.free()


def test_16bit_column_stream_is_bit_identical():
    """The whole-operator kernels read the columns as 16-bit window codes where every row range fits four windows of
    16 K columns (tune key 19 switches back to 32-bit columns): same bits either way, on a banded stencil (three
    windows per range), a wide random operator (falls back to 32-bit: more than four windows) and BSR(1,1)."""
    import scipy.sparse as sp
    from tools.problems import poisson_csr
    rng = np.random.RandomState(5)
    nw = 70000                                             # wide random operator: 12 scattered columns per row
    cols = rng.randint(0, nw, size=(nw, 12)).astype(np.int32)
    wide = sp.csr_array((rng.rand(nw * 12), cols.ravel(), np.arange(0, nw * 12 + 1, 12, dtype=np.int32)), shape=(nw, nw))
    wide.sum_duplicates()
    ops = [poisson_csr((40, 40, 40)), wide, sp.bsr_array(poisson_csr((64, 64)), blocksize=(1, 1))]
    for A in ops:
        n = A.shape[0]
        x, b = rng.rand(n), rng.rand(n)
        dA = DeviceMatrix(sparse_op(A))
        dx, db = capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(b)
        out = {}
        for flag in (1, 0):
            dA.tune(idx16=flag)
            dy = capi.DeviceArray(n, np.float64)
            dA.spmv(capi.SPMV_RESID, dx, dy, b=db)
            dj = capi.DeviceArray.from_host(x)
            dw = capi.DeviceArray(n, np.float64)
            dA.jacobi(dj, db, dw, 0.8, iterations=2)
            out[flag] = (dy.download(), dj.download())
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        assert np.array_equal(out[1][0], b - sp.csr_array(A) @ x)
        dA.free()
