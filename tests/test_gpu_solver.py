"""GPU parity, solver level: DeviceMultilevelSolver vs the real reference's outputs on
identical hierarchies (tests/golden/hier_*.npz, built by the reference) and vs the oracle.

Bars (written into each test): f64 residual norms within 1e-10 relative of the reference
for every cycle under the reference's own protocol (b = 0, x0 = rand,
docs/paper/example.py:11-14); for random b: |r_gpu - r_ref| <= 1e-10 * ||r_0|| (the norm of
b - A x cancels digits as the solve converges); iterates agree to 1e-12
relative.  f32: 2e-4 relative."""
import numpy as np
import pytest

from conftest import golden_hierarchies
from pyamg_amd import DeviceMultilevelSolver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_hierarchies())
@pytest.mark.parametrize("graph", [True, False])
def test_solve_matches_reference(load_hier, name, graph):
    spec, ex = load_hier(name)
    k, cycle = int(ex["k"]), str(ex["cycle"])
    f32 = spec.dtype == np.float32
    dml = DeviceMultilevelSolver(spec, graph=graph)
    # reference protocol: b = 0, x0 = rand -> relative match of every residual norm
    resz = []
    xz = dml.solve(np.zeros_like(ex["x0z"]), x0=ex["x0z"], tol=1e-30, maxiter=k, cycle=cycle, residuals=resz)
    resz = np.array(resz)
    assert resz.shape == ex["resz"].shape
    rel = np.max(np.abs(resz - ex["resz"]) / ex["resz"])
    assert rel <= (2e-4 if f32 else 1e-10), rel
    # random rhs: absolute match scaled by the initial residual
    res = []
    x, info = dml.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, cycle=cycle, residuals=res, return_info=True)
    res = np.array(res)
    assert info == k and len(res) == k + 1
    assert np.max(np.abs(res - ex["res"])) <= (2e-4 if f32 else 1e-10) * ex["res"][0]
    assert np.linalg.norm(x - ex["x"]) <= (2e-4 if f32 else 1e-12) * np.linalg.norm(ex["x"])
    assert x.dtype == spec.dtype and x.shape == (spec.levels[0].A.shape[0],)
    assert np.linalg.norm(xz - ex["xz"]) <= (2e-4 if f32 else 1e-12) * max(np.linalg.norm(ex["xz"]), 1e-300) + 1e-300


@pytest.mark.parametrize("name", ["sa2d_cheby", "sa2d_jacobi", "rs2d_jacobi", "rs3d_jacobi_f32", "sa2d_richardson_W", "sa2d_coarse_jacobi", "sa2d_jacobi_AMLI"])
def test_renumbered_interior_levels_give_the_same_iterates(load_hier, name, monkeypatch):
    """renumber=True moves the unknowns of the interior levels (pyamg_amd/renumber.py): rows moved, columns renamed, row sums in stored
    order -- the level-0 iterates are the SAME BITS as with the reference's numbering (AMLI adds inner products over level vectors, whose
    order of summation moves: 1e-12 there), and both match the reference's recorded run."""
    spec, ex = load_hier(name)
    k, cycle = int(ex["k"]), str(ex["cycle"])
    f32 = spec.dtype == np.float32
    monkeypatch.setenv("PAMG_RENUMBER_MIN_ROWS", "8")
    on = DeviceMultilevelSolver(spec, renumber=True)
    off = DeviceMultilevelSolver(spec, renumber=False)
    assert on.renumbered and off.renumbered == []
    assert on.levels[1].A is spec.levels[1].A                     # what the caller sees keeps the reference's numbering
    for cyc in ([cycle] if cycle == "AMLI" else [cycle, "W", "F"]):
        r1, r2 = [], []
        x1 = on.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, cycle=cyc, residuals=r1)
        x2 = off.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, cycle=cyc, residuals=r2)
        if cyc == "AMLI":
            assert np.linalg.norm(x1 - x2) <= 1e-12 * np.linalg.norm(x2)
        else:
            assert np.array_equal(x1, x2) and np.array_equal(r1, r2), cyc
    res = []
    x = on.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, cycle=cycle, residuals=res)
    assert np.max(np.abs(np.array(res) - ex["res"])) <= (2e-4 if f32 else 1e-10) * ex["res"][0]
    assert np.linalg.norm(x - ex["x"]) <= (2e-4 if f32 else 1e-12) * np.linalg.norm(ex["x"])
    # Krylov acceleration around the renumbered hierarchy
    if cycle != "AMLI":
        xa = on.solve(ex["b"], tol=1e-10, accel="cg" if name.startswith("sa2d") else "gmres")
        xb = off.solve(ex["b"], tol=1e-10, accel="cg" if name.startswith("sa2d") else "gmres")
        assert np.array_equal(xa, xb)


def test_solve_api_conventions(load_hier):
    """Return/shape/info/callback conventions of multilevel.py:398-582."""
    spec, ex = load_hier("sa2d_gs")
    dml = DeviceMultilevelSolver(spec)
    n = spec.levels[0].A.shape[0]
    b = ex["b"]
    # converges -> info == 0, len(residuals) == iterations + 1, (n,1) input ravelled
    res = []
    x, info = dml.solve(b.reshape(-1, 1), tol=1e-8, residuals=res, return_info=True)
    assert info == 0 and x.shape == (n,)
    assert res[-1] < 1e-8 * np.linalg.norm(b) <= res[-2]
    # callback sees every iterate; same answer as without callback
    seen = []
    x2 = dml.solve(b, tol=1e-8, callback=lambda xk: seen.append(xk.copy()))
    assert len(seen) == len(res) - 1 and np.array_equal(seen[-1], x2) and np.array_equal(x2, x)
    # maxiter hit -> info == iteration count
    _, info = dml.solve(b, tol=1e-30, maxiter=3, return_info=True)
    assert info == 3
    # aspreconditioner == exactly one cycle from zero (multilevel.py:390-396)
    M = dml.aspreconditioner()
    z1 = M @ b
    z2 = dml.solve(b, maxiter=1, tol=1e-12)
    assert np.array_equal(z1, z2) and (M @ b.reshape(-1, 1)).shape == (n, 1)
    with pytest.raises(ValueError):
        dml.solve(b, cycle="AMLI", accel="cg")          # multilevel.py:488-490
    with pytest.raises(TypeError):
        dml.solve(b, cycle="Q")


def test_krylov_acceleration(load_hier):
    """accel= path (multilevel.py:479-535): host CG/GMRES around the device preconditioner."""
    import scipy.sparse.linalg as sla
    spec, ex = load_hier("sa2d_gs")
    dml = DeviceMultilevelSolver(spec)
    A = spec.levels[0].A.to_scipy()
    b = ex["b"]
    res = []
    x = dml.solve(b, tol=1e-10, accel="cg", residuals=res)
    assert np.linalg.norm(b - A @ x) <= 1e-8 * np.linalg.norm(b) and len(res) > 1
    x, info = sla.gmres(A, b, M=dml.aspreconditioner("W"), rtol=1e-10)
    assert info == 0 and np.linalg.norm(b - A @ x) <= 1e-8 * np.linalg.norm(b)


def test_against_oracle_cycles(load_hier):
    """V/W/F cycles vs the oracle's restatement of __solve on the same hierarchy."""
    from oracle import oracle as orc
    spec, ex = load_hier("sa2d_cheby")
    dml = DeviceMultilevelSolver(spec)
    osol = orc.OracleSolver(spec)
    for cyc, cpl in (("V", 1), ("W", 1), ("F", 1), ("F", 2)):
        r1, r2 = [], []
        x1 = dml.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=4, cycle=cyc, cycles_per_level=cpl, residuals=r1)
        x2 = osol.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=4, cycle=cyc, cycles_per_level=cpl, residuals=r2)
        assert np.max(np.abs(np.array(r1) - np.array(r2))) <= 1e-10 * r2[0], cyc
        assert np.linalg.norm(x1 - x2) <= 1e-12 * np.linalg.norm(x2), cyc


def test_with_live_reference_if_available():
    """When oracle/_ref travelled to this box: build a fresh hierarchy with the real
    reference and compare the first 10 residual norms (1e-10 relative, b = 0 protocol)."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present on this box")
    import pyamg
    A = pyamg.gallery.poisson((64, 64, 16), format="csr")
    np.random.seed(77)
    for kw in (dict(), dict(presmoother=("jacobi", {"omega": 4 / 3}), postsmoother=("jacobi", {"omega": 4 / 3})),
               dict(presmoother=("chebyshev", {"degree": 3}), postsmoother=("chebyshev", {"degree": 3}))):
        ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, **kw)
        dml = DeviceMultilevelSolver(ml)
        x0 = np.random.rand(A.shape[0])
        b = np.zeros(A.shape[0])
        r_ref, r_gpu = [], []
        ml.solve(b, x0=x0, tol=1e-30, maxiter=10, residuals=r_ref)
        dml.solve(b, x0=x0, tol=1e-30, maxiter=10, residuals=r_gpu)
        r_ref, r_gpu = np.array(r_ref), np.array(r_gpu)
        assert np.max(np.abs(r_gpu - r_ref) / r_ref) <= 1e-10


def _reference_pcg(A, b, M, tol, maxiter):
    """NumPy restatement of pyamg/krylov/_cg.py:98-198 (criteria 'rr') for the test."""
    x = np.zeros_like(b)
    r = b - A @ x
    z = M(r)
    p = z.copy()
    rz = np.inner(r, z)
    res = [np.linalg.norm(r)]
    normb = np.linalg.norm(b) or 1.0
    it = 0
    while True:
        Ap = A @ p
        rz_old = rz
        alpha = rz / np.inner(Ap, p)
        x += alpha * p
        if np.mod(it, 8) and it > 0:
            r -= alpha * Ap
        else:
            r = b - A @ x
        z = M(r)
        rz = np.inner(r, z)
        p *= rz / rz_old
        p += z
        it += 1
        res.append(np.linalg.norm(r))
        if res[-1] < tol * normb or it == maxiter:
            return x, res


def test_device_pcg_matches_reference_algorithm(load_hier):
    """solve(accel='cg') runs CG on the device; compare with the reference's algorithm driven by
    the oracle's V-cycle as preconditioner (f64: 1e-9 relative on the first residual norms --
    CG amplifies the last-bit differences of the dot products)."""
    from oracle import oracle as orc
    spec, ex = load_hier("sa2d_gs")
    dml = DeviceMultilevelSolver(spec)
    osol = orc.OracleSolver(spec)
    A = spec.levels[0].A.to_scipy()
    b = ex["b"]

    def M(r):
        return osol.solve(r, maxiter=1, tol=1e-12)

    xo, ro = _reference_pcg(A, b, M, 1e-10, 12)
    res = []
    x, info = dml.solve(b, tol=1e-10, maxiter=12, accel="cg", residuals=res, return_info=True)
    m = min(len(res), len(ro), 9)
    assert np.max(np.abs(np.array(res[:m]) - np.array(ro[:m])) / np.array(ro[:m])) <= 1e-9
    assert np.linalg.norm(b - A @ x) <= 1e-8 * np.linalg.norm(b)
    # converged run: info == 0 and the last residual is below tol*||b||
    res = []
    x, info = dml.solve(b, tol=1e-8, maxiter=50, accel="cg", residuals=res, return_info=True)
    assert info == 0 and res[-1] < 1e-8 * np.linalg.norm(b) <= res[-2]


def test_device_pcg_matches_reference_history():
    """solve(accel='cg') on the device against the reference's OWN MultilevelSolver.solve(accel='cg') histories on
    the committed hierarchies (tests/golden/accel_cg.npz, make_golden.py --accel-only): same list length and info,
    every residual norm within 1e-10 ||r0||, solution within 1e-9 relative (SA 2-D / 3-D symmetric GS, BSR block GS)."""
    from conftest import GOLDEN
    from pyamg_amd.hierarchy import load_spec
    z = np.load(GOLDEN / "accel_cg.npz")
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) >= 3
    for name in names:
        spec, _ = load_spec(GOLDEN / f"hier_{name}.npz")
        dml = DeviceMultilevelSolver(spec)
        b = z[f"{name}.b"]
        cyc = str(z[f"{name}.cycle"])
        for tag in ("a", "b"):
            res = []
            x, info = dml.solve(b, tol=float(z[f"{name}.{tag}.tol"]), maxiter=int(z[f"{name}.{tag}.maxiter"]), cycle=cyc,
                                accel="cg", residuals=res, return_info=True)
            ref = z[f"{name}.{tag}.res"]
            assert len(res) == len(ref) and info == int(z[f"{name}.{tag}.info"]), (name, tag, len(res), len(ref), info)
            assert np.max(np.abs(np.array(res) - ref)) <= 1e-10 * ref[0], (name, tag)
            xr = z[f"{name}.{tag}.x"]
            assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr), (name, tag)
        dml.free()


def test_change_solve_matrix_and_fgmres_amli_with_live_reference():
    """change_solve_matrix re-ships the hierarchy (multilevel.py:320-337); AMLI cycle as an
    FGMRES preconditioner (the only accelerator the reference allows with AMLI, :488-490)."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present on this box")
    import pyamg
    A = pyamg.gallery.poisson((40, 40), format="csr")
    np.random.seed(5)
    jac = ("jacobi", {"omega": 4 / 3})
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=jac, postsmoother=jac)
    dml = DeviceMultilevelSolver(ml)
    b = np.random.rand(A.shape[0])
    r_ref, r_gpu = [], []
    x_ref = ml.solve(b, tol=1e-10, maxiter=15, cycle="AMLI", accel="fgmres", residuals=r_ref)
    x_gpu = dml.solve(b, tol=1e-10, maxiter=15, cycle="AMLI", accel="fgmres", residuals=r_gpu)
    m = min(len(r_ref), len(r_gpu), 6)
    assert np.max(np.abs(np.array(r_gpu[:m]) - np.array(r_ref[:m]))) <= 1e-8 * r_ref[0]
    assert np.linalg.norm(b - A @ x_gpu) <= 1e-8 * np.linalg.norm(b)
    A2 = (A * 2.0).tocsr()
    dml.change_solve_matrix(A2)
    r_ref, r_gpu = [], []
    ml.solve(b, tol=1e-30, maxiter=4, residuals=r_ref)
    dml.solve(b, tol=1e-30, maxiter=4, residuals=r_gpu)
    assert np.max(np.abs(np.array(r_gpu) - np.array(r_ref))) <= 1e-10 * r_ref[0]


@pytest.mark.parametrize("name", ["rs2d_nonsym_gsne", "rs2d_nonsym_jacobine"])
def test_ne_smoother_on_an_unsorted_bsr_level(load_hier, name):
    """On a BSR(1,1) level whose rows are stored unsorted the reference smooths with the sorted lvl.Acsr while the
    cycle's own products keep the stored order (smoothing.py setup_*_ne): the spec then carries the sorted copy
    (SmootherSpec.Ar) for the smoother.  Built here from a committed hierarchy by storing level 1 as BSR(1,1) with
    every row reversed; the device must agree with the oracle, which applies the same rule."""
    import copy
    from oracle import oracle as orc
    from pyamg_amd.hierarchy import SparseOp
    spec, ex = load_hier(name)
    spec = copy.deepcopy(spec)
    L = spec.levels[1]
    A = L.A
    assert A.blocksize == (1, 1)
    idx, dat = A.indices.copy(), np.ravel(A.data).copy()
    for i in range(A.shape[0]):
        p0, p1 = int(A.indptr[i]), int(A.indptr[i + 1])
        idx[p0:p1] = idx[p0:p1][::-1]
        dat[p0:p1] = dat[p0:p1][::-1]
    sorted_copy = SparseOp("csr", A.shape, (1, 1), A.indptr.copy(), A.indices.copy(), np.ravel(A.data).copy(), "csr")
    L.A = SparseOp("bsr", A.shape, (1, 1), A.indptr.copy(), idx, dat, "bsr")
    for sm in (L.pre, L.post):
        sm.Ar = sorted_copy
    dml = DeviceMultilevelSolver(spec)
    r_gpu, r_orc = [], []
    x = dml.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=5, residuals=r_gpu)
    xo = orc.OracleSolver(spec).solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=5, residuals=r_orc)
    dml.free()
    assert np.max(np.abs(np.array(r_gpu) - np.array(r_orc))) <= 1e-10 * r_orc[0]
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)


def test_midsize_symmetric_gs_against_the_live_reference():
    """3-D Poisson 96^3 (885K rows) SA, symmetric Gauss-Seidel: big enough that the fine level runs the tiled sweep on
    pencil tiles across all XCDs, level 1 the multi-XCD granular sweep and the coarse levels the single-workgroup /
    single-tile forms.  The reference's own protocol (b = 0, x0 = rand; docs/paper/example.py:11-14): every residual
    norm of 10 V-cycles within 1e-10 RELATIVE of the reference's, the iterate within 1e-12; graph replay and eager
    launches agree bit for bit."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present on this box")
    import pyamg
    A = pyamg.gallery.poisson((96, 96, 96), format="csr")
    np.random.seed(11)
    gs = ("gauss_seidel", {"sweep": "symmetric"})
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=gs, postsmoother=gs)
    np.random.seed(2022)
    x0 = np.random.rand(A.shape[0])
    b = np.zeros(A.shape[0])
    r_ref = []
    x_ref = ml.solve(b, x0=x0, tol=1e-30, maxiter=10, residuals=r_ref)
    for order in ("fast", "exact"):
        outs = []
        for graph in (True, False):
            dml = DeviceMultilevelSolver(ml, graph=graph, order=order)
            r_gpu = []
            outs.append(dml.solve(b, x0=x0, tol=1e-30, maxiter=10, residuals=r_gpu))
            if graph:
                tiles = [dA.tile_info(0)["tiles"] for dA in dml.A[:-1]]
                lines = [dA.line_info(0)["lines"] for dA in dml.A[:-1]]
                lanes = [dA.lane_info(0)["groups"] for dA in dml.A[:-1]]
                merged = [dA.lanem_info(0) for dA in dml.A[:-1]]
            dml.free()
            r_ref_a, r_gpu_a = np.array(r_ref), np.array(r_gpu)
            assert len(r_gpu_a) == len(r_ref_a) == 11
            assert np.max(np.abs(r_gpu_a - r_ref_a) / r_ref_a) <= 1e-10, order
            assert np.linalg.norm(outs[-1] - x_ref) <= 1e-12 * np.linalg.norm(x_ref), order
        assert np.array_equal(outs[0], outs[1]), order
        if order == "exact":
            assert tiles[0] > 1 and not any(lines) and not any(lanes) and not any(m["rows"] for m in merged)     # the fine level runs the tiled sweep
        else:
            # fine level: one line per grid line; SA level 1: the MERGED lane form (round 6: dependency levels eliminated into super-levels)
            assert lines[0] == 96 * 96 and merged[1]["rows"] > 0 and merged[1]["super_levels"] * 2 < merged[1]["dependency_levels"], merged[1]


@pytest.mark.parametrize("case", ["poisson3d_128", "poisson2d_2000"])
def test_long_dependency_chains_against_the_live_reference(case):
    """Order-exact symmetric Gauss-Seidel where the dependency chains are LONG, against the live reference (b = 0,
    x0 = rand, every residual norm within 1e-10 relative): 3-D Poisson 128^3 SA (2.1 M rows; its SA level 1 -- 263 K
    rows of ~30 entries, 613 dependency levels -- run by the default scheduler AND forced onto the multi-XCD granular
    sweep the 256^3 hierarchy uses there), and 2-D Poisson 2000^2 SA (4 M rows: 3 999 dependency levels on the fine
    level, ~2 000 on level 1).  A scheduler change that only breaks on deep schedules is caught here, not only by bench.py."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present on this box")
    import pyamg
    grid = (128, 128, 128) if case == "poisson3d_128" else (2000, 2000)
    A = pyamg.gallery.poisson(grid, format="csr")
    np.random.seed(5)
    gs = ("gauss_seidel", {"sweep": "symmetric"})
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=gs, postsmoother=gs)
    np.random.seed(2022)
    x0 = np.random.rand(A.shape[0])
    b = np.zeros(A.shape[0])
    k = 3
    r_ref = []
    x_ref = ml.solve(b, x0=x0, tol=1e-30, maxiter=k, residuals=r_ref)
    r_ref = np.array(r_ref)
    # the fast order (lane-parallel row sums, the default) and the order-exact schedulers: all within 1e-10 of the
    # reference; the exact ones identical to each other
    tunes = [("fast", None), ("exact", None)]
    if case == "poisson3d_128":
        tunes.append(("exact", lambda i: {"gs_mode": 2, "gran_xcd": 2} if i == 1 else None))
    outs = []
    for order, tune in tunes:
        dml = DeviceMultilevelSolver(ml, level_tune=tune, order=order)
        if order == "fast":
            assert dml.A[1].lanem_info(0)["rows"] > 0 or dml.A[1].lane_info(0)["groups"] > 0      # SA level 1 runs the lane form (merged where rows are long enough)
        depth = [dA.info()["gs_levels_fwd"] for dA in dml.A[:-1]]
        r_gpu = []
        outs.append(dml.solve(b, x0=x0, tol=1e-30, maxiter=k, residuals=r_gpu))
        assert not any(dA.flow_error() for dA in dml.A)
        dml.free()
        r_gpu = np.array(r_gpu)
        assert len(r_gpu) == len(r_ref) == k + 1
        assert np.max(np.abs(r_gpu - r_ref) / r_ref) <= 1e-10, (case, np.max(np.abs(r_gpu - r_ref) / r_ref))
        assert np.linalg.norm(outs[-1] - x_ref) <= 1e-12 * np.linalg.norm(x_ref)
        assert max(depth) >= (2000 if case == "poisson2d_2000" else 600), depth
    for o in outs[2:]:
        assert np.array_equal(o, outs[1])                 # exact schedulers differ in speed only


@pytest.mark.parametrize("hier", ["sa3d_gs", "rs2d_nonsym_gsnr", "rs2d_nonsym_gsne", "sa2d_schwarz"])
def test_sweep_timeout_falls_back_to_level_launches(hier):
    """a persistent sweep that reports PAMG_E_TIMEOUT (not all of its workgroups were running -- forced here through the
    PAMG_FORCE_TIMEOUT test hook): solve() switches every order-exact sweep to one launch per dependency level, runs the
    solve again from the initial guess and returns the reference's answer; the switch is reported in stats() and by ONE
    RuntimeWarning.  The normal-equation hierarchies: their Kaczmarz lane sweeps run on the smoother's OWN operators
    (A^T / the row-sorted twin), which the fallback must switch to per-level launches too; so must the persistent Schwarz sweep (round 6)"""
    import subprocess
    import sys
    from conftest import ROOT
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from pyamg_amd import DeviceMultilevelSolver\n"
        "from pyamg_amd.hierarchy import load_spec\n"
        f"spec, ex = load_spec({str(ROOT / 'tests' / 'golden' / ('hier_' + hier + '.npz'))!r})\n"
        "dml = DeviceMultilevelSolver(spec)\n"
        "r = []\n"
        "x = dml.solve(ex['b'], x0=ex['x0'], tol=1e-30, maxiter=int(ex['k']), residuals=r)\n"
        "st = dml.stats()\n"
        "assert st['sweep_timeouts_recovered'] == 1, st\n"
        "assert np.max(np.abs(np.array(r) - ex['res'])) <= 1e-10 * ex['res'][0]\n"
        "assert np.linalg.norm(x - ex['x']) <= 1e-12 * np.linalg.norm(ex['x'])\n"
        "r2 = []\n"
        "x2 = dml.solve(ex['b'], x0=ex['x0'], tol=1e-30, maxiter=int(ex['k']), residuals=r2)\n"
        "assert np.array_equal(x2, x) and dml.stats()['sweep_timeouts_recovered'] == 1\n"
        "print('fallback ok')\n")
    import os
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PAMG_FORCE_TIMEOUT="1"))
    assert r.returncode == 0 and "fallback ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert r.stderr.count("a persistent Gauss-Seidel sweep timed out") == 1, r.stderr[-2000:]      # said once, not silently slower


def test_device_fgmres_matches_reference():
    """solve(accel='fgmres') runs flexible GMRES on the device (pamg_solver_fgmres); compared with the
    reference's own MultilevelSolver.solve(accel='fgmres') on the committed hierarchies
    (tests/golden/accel_fgmres.npz, generated by make_golden.py --accel-only): same list length, same
    info, residual norms within 1e-10 ||r0||, solution within 1e-9 relative.  Covers V and AMLI cycles
    (the only accelerator the reference allows with AMLI), a non-symmetric operator, BSR block
    Gauss-Seidel and a run that hits maxiter."""
    from conftest import GOLDEN
    from pyamg_amd.hierarchy import load_spec
    z = np.load(GOLDEN / "accel_fgmres.npz")
    names = sorted({k.split(".")[0] for k in z.files})      # keys: <hierarchy>.[gmres.]<a|b>.<res|x|info|...>
    assert len(names) >= 5
    for name in names:
        spec, _ = load_spec(GOLDEN / f"hier_{name}.npz")
        dml = DeviceMultilevelSolver(spec)
        b = z[f"{name}.b"]
        cyc = str(z[f"{name}.cycle"])
        for tag in ("a", "b"):
            res = []
            x, info = dml.solve(b, tol=float(z[f"{name}.{tag}.tol"]), maxiter=int(z[f"{name}.{tag}.maxiter"]), cycle=cyc,
                                accel="fgmres", residuals=res, return_info=True)
            ref = z[f"{name}.{tag}.res"]
            assert len(res) == len(ref) and info == int(z[f"{name}.{tag}.info"]), (name, tag, len(res), len(ref), info)
            assert np.max(np.abs(np.array(res) - ref)) <= 1e-10 * ref[0], (name, tag)
            xr = z[f"{name}.{tag}.x"]
            assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr), (name, tag)
            if f"{name}.gmres.{tag}.res" not in z.files:
                continue
            # the reference's default GMRES (left-preconditioned Householder): pamg_solver_gmres
            res = []
            x, info = dml.solve(b, tol=float(z[f"{name}.{tag}.tol"]), maxiter=int(z[f"{name}.{tag}.maxiter"]), cycle=cyc,
                                accel="gmres", residuals=res, return_info=True)
            ref = z[f"{name}.gmres.{tag}.res"]
            assert len(res) == len(ref) and info == int(z[f"{name}.gmres.{tag}.info"]), (name, tag, "gmres", len(res), len(ref), info)
            assert np.max(np.abs(np.array(res) - ref)) <= 1e-10 * ref[0], (name, tag, "gmres")
            xr = z[f"{name}.gmres.{tag}.x"]
            assert np.linalg.norm(x - xr) <= 1e-9 * np.linalg.norm(xr), (name, tag, "gmres")
        dml.free()


@pytest.mark.parametrize("coarse", ["bicgstab", "callable"])       # (cgs / qmr / bicg / minres are SciPy's: with the SciPy installed here the reference itself fails on their tol= argument)
def test_host_coarse_solvers_against_the_live_reference(coarse):
    """coarse_solver = a Krylov name other than 'cg' / 'gmres', or a callable (multilevel.py:752-762, 786-788): the coarse
    right-hand side goes to the caller's OWN solver object on the host inside the device cycle
    (pamg_solver_set_coarse_host); residual histories within 1e-10 of the reference's, V and W cycles."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present on this box")
    import pyamg
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    A = pyamg.gallery.poisson((40, 40), format="csr")
    cs = (lambda A_, b_: sla.spsolve(sp.csc_array(A_), b_)) if coarse == "callable" else coarse
    np.random.seed(7)
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=40, coarse_solver=cs)
    b = np.random.rand(A.shape[0])
    dml = DeviceMultilevelSolver(ml)
    assert dml.spec.coarse_kind == "host"
    for cyc in ("V", "W"):
        r_ref, r_gpu = [], []
        x_ref = ml.solve(b, tol=1e-30, maxiter=6, cycle=cyc, residuals=r_ref)
        x = dml.solve(b, tol=1e-30, maxiter=6, cycle=cyc, residuals=r_gpu)
        r_ref, r_gpu = np.array(r_ref), np.array(r_gpu)
        assert len(r_gpu) == len(r_ref) == 7
        assert np.max(np.abs(r_gpu - r_ref)) <= 1e-10 * r_ref[0], (coarse, cyc)
        assert np.linalg.norm(x - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
    dml.free()
