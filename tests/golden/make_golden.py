#!/usr/bin/env python3
"""Generate tests/golden/*.npz and known_answers.json from the REAL reference.

Run in the build container (needs /root/reference, built into oracle/_ref by
oracle/build_ref.py):   python tests/golden/make_golden.py

Every ``hier_*.npz`` holds one hierarchy built by the reference (levels A/P/R, smoother
parameters read back from the constructed solver, coarse operator), seeded inputs and the
reference's own outputs for them:
    b, x0          inputs of the random-rhs run          -> res, x (after k cycles)
    x0z            input of the reference's protocol run (b = 0, x0 = rand,
                   docs/paper/example.py:11-14)           -> resz
``kernels.npz`` holds kernel-level input/output pairs produced by the reference's
``amg_core`` / ``relaxation`` functions and SciPy's ``A @ x``.
``known_answers.json`` restates the known answers the reference's own tests pin
(file:line cited per entry).
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

import oracle.refimport  # noqa: E402,F401
import pyamg  # noqa: E402
from pyamg.relaxation import relaxation as rr  # noqa: E402
from pyamg_amd import hierarchy  # noqa: E402

SEED = 20260924


ONLY = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]     # regenerate selected hierarchies only


ACCEL_ONLY = "--accel-only" in sys.argv      # only (re)generate accel_fgmres.npz
ACCEL_CASES = ("sa2d_gs", "sa2d_jacobi_AMLI", "rs2d_nonsym_gs", "el2d_blockgs", "sa3d_gs", "air2d_fcjacobi",
               "bb2d_nonsym_gsnr")
ACCEL = {}
ACCEL_CG = {}
CG_CASES = ("sa2d_gs", "sa3d_gs", "el2d_blockgs")     # SPD operator + symmetric smoothing: solve(accel='cg') is legitimate


def accel_case(name, ml, cycle):
    """MultilevelSolver.solve(accel='fgmres') of the reference (multilevel.py:479-535 ->
    krylov/_fgmres.py) on the same hierarchy: residual history, solution, info."""
    n = ml.levels[0].A.shape[0]
    b = np.random.RandomState(SEED + 7).rand(n)
    for tag, kw in (("a", dict(tol=1e-10, maxiter=12)), ("b", dict(tol=1e-6, maxiter=40))):
        res = []
        x, info = ml.solve(b, cycle=cycle, accel="fgmres", residuals=res, return_info=True, **kw)
        ACCEL[f"{name}.{tag}.res"] = np.array(res)
        ACCEL[f"{name}.{tag}.x"] = x
        ACCEL[f"{name}.{tag}.info"] = np.array(info)
        ACCEL[f"{name}.{tag}.tol"] = np.array(kw["tol"])
        ACCEL[f"{name}.{tag}.maxiter"] = np.array(kw["maxiter"])
    if cycle != "AMLI":        # the reference allows AMLI cycles only under fgmres (multilevel.py:488-490)
        for tag, kw in (("a", dict(tol=1e-10, maxiter=12)), ("b", dict(tol=1e-6, maxiter=40))):
            res = []
            x, info = ml.solve(b, cycle=cycle, accel="gmres", residuals=res, return_info=True, **kw)
            ACCEL[f"{name}.gmres.{tag}.res"] = np.array(res)
            ACCEL[f"{name}.gmres.{tag}.x"] = x
            ACCEL[f"{name}.gmres.{tag}.info"] = np.array(info)
    if name in CG_CASES:
        # MultilevelSolver.solve(accel='cg') of the reference (multilevel.py:479-535 -> krylov/_cg.py)
        for tag, kw in (("a", dict(tol=1e-10, maxiter=12)), ("b", dict(tol=1e-8, maxiter=50))):
            res = []
            x, info = ml.solve(b, cycle=cycle, accel="cg", residuals=res, return_info=True, **kw)
            ACCEL_CG[f"{name}.{tag}.res"] = np.array(res)
            ACCEL_CG[f"{name}.{tag}.x"] = x
            ACCEL_CG[f"{name}.{tag}.info"] = np.array(info)
            ACCEL_CG[f"{name}.{tag}.tol"] = np.array(kw["tol"])
            ACCEL_CG[f"{name}.{tag}.maxiter"] = np.array(kw["maxiter"])
        ACCEL_CG[f"{name}.b"] = b
        ACCEL_CG[f"{name}.cycle"] = np.array(cycle)
    ACCEL[f"{name}.b"] = b
    ACCEL[f"{name}.cycle"] = np.array(cycle)
    print(f"accel fgmres {name}: lens {len(ACCEL[name + '.a.res'])}/{len(ACCEL[name + '.b.res'])} info {ACCEL[name + '.a.info']}/{ACCEL[name + '.b.info']}")


def hier(name, ml, k=8, cycle="V"):
    if ONLY and name not in ONLY:
        return
    if name in ACCEL_CASES:
        if ACCEL_ONLY:
            # the hierarchy must be the committed one, array for array
            ref, _ = hierarchy.load_spec(HERE / f"hier_{name}.npz")
            new = hierarchy.extract(ml)
            assert len(ref.levels) == len(new.levels)
            for La, Lb in zip(ref.levels, new.levels):
                for oa, ob in ((La.A, Lb.A), (La.P, Lb.P), (La.R, Lb.R)):
                    if oa is not None:
                        assert np.array_equal(oa.data, ob.data) and np.array_equal(oa.indices, ob.indices), name
        accel_case(name, ml, cycle)
    if ACCEL_ONLY:
        return
    spec = hierarchy.extract(ml)
    n = ml.levels[0].A.shape[0]
    rng = np.random.RandomState(SEED)
    dt = ml.levels[0].A.dtype
    b = rng.rand(n).astype(dt)
    x0 = rng.rand(n).astype(dt)
    x0z = rng.rand(n).astype(dt)
    res, resz = [], []
    x = ml.solve(b, x0=x0, tol=1e-30, maxiter=k, cycle=cycle, residuals=res)
    xz = ml.solve(np.zeros(n, dtype=dt), x0=x0z, tol=1e-30, maxiter=k, cycle=cycle, residuals=resz)
    hierarchy.save_spec(HERE / f"hier_{name}.npz", spec, b=b, x0=x0, x0z=x0z, res=np.array(res), x=x,
                        resz=np.array(resz), xz=xz, k=k, cycle=np.array(cycle))
    print(f"hier_{name}: levels={len(ml.levels)} n={n} conv={res[-1] / res[0]:.2e} convz={resz[-1] / resz[0]:.2e}")


def make_hierarchies():
    A = pyamg.gallery.poisson((40, 40), format="csr")
    jac = ("jacobi", {"omega": 4.0 / 3.0})
    np.random.seed(SEED)
    hier("sa2d_gs", pyamg.smoothed_aggregation_solver(A, max_coarse=10))
    np.random.seed(SEED)
    hier("sa2d_jacobi", pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=jac, postsmoother=jac))
    np.random.seed(SEED)
    ch = ("chebyshev", {"degree": 3, "iterations": 1})
    hier("sa2d_cheby", pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=ch, postsmoother=ch))
    np.random.seed(SEED)
    hier("sa2d_sor", pyamg.smoothed_aggregation_solver(
        A, max_coarse=10, presmoother=("sor", {"omega": 1.2, "sweep": "forward"}),
        postsmoother=("sor", {"omega": 1.2, "sweep": "backward"})))
    np.random.seed(SEED)
    hier("sa2d_richardson_W", pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother="richardson",
                                                                postsmoother="richardson"), cycle="W")
    # relaxation methods as the coarsest-level solver (multilevel.py:765-782): sweeps from x = 0 on a 50-100 unknown level
    np.random.seed(SEED)
    hier("sa2d_coarse_gs", pyamg.smoothed_aggregation_solver(A, max_coarse=60, coarse_solver="gauss_seidel"))
    np.random.seed(SEED)
    hier("sa2d_coarse_jacobi", pyamg.smoothed_aggregation_solver(A, max_coarse=60, presmoother=jac, postsmoother=jac,
                                                                 coarse_solver=("jacobi", {"iterations": 4})))
    np.random.seed(SEED)
    hier("sa2d_coarse_cheby", pyamg.smoothed_aggregation_solver(A, max_coarse=60, coarse_solver=("chebyshev", {"degree": 4, "iterations": 2})))
    # Krylov methods as smoothers (smoothing.py:794-830) and as coarse solvers (multilevel.py:752-762)
    for tag, meth in (("cg", ("cg", {"maxiter": 2})), ("gmres", ("gmres", {"maxiter": 3}))):
        np.random.seed(SEED)
        hier(f"sa2d_{tag}", pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=meth, postsmoother=meth))
    np.random.seed(SEED)
    hier("sa2d_coarse_cg", pyamg.smoothed_aggregation_solver(A, max_coarse=60, coarse_solver="cg"))
    np.random.seed(SEED)
    hier("sa2d_coarse_gmres", pyamg.smoothed_aggregation_solver(A, max_coarse=60, coarse_solver="gmres"))
    np.random.seed(SEED)
    hier("rs2d_jacobi", pyamg.ruge_stuben_solver(A, max_coarse=10, presmoother=jac, postsmoother=jac))
    np.random.seed(SEED)
    hier("rs2d_gs_F", pyamg.ruge_stuben_solver(A, max_coarse=10), cycle="F")
    np.random.seed(SEED)
    hier("sa2d_jacobi_AMLI", pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=jac, postsmoother=jac),
         cycle="AMLI", k=6)
    A3 = pyamg.gallery.poisson((12, 12, 12), format="csr")
    np.random.seed(SEED)
    hier("sa3d_gs", pyamg.smoothed_aggregation_solver(A3, max_coarse=10,
                                                      presmoother=("gauss_seidel", {"sweep": "symmetric"}),
                                                      postsmoother=("gauss_seidel", {"sweep": "symmetric"})))
    np.random.seed(SEED)
    # float32: Ruge-Stuben keeps the whole hierarchy in float32 (SA promotes P and the coarse levels
    # to float64 -- a mixed-precision hierarchy, which the device path rejects explicitly)
    A3f = A3.astype(np.float32)
    hier("rs3d_jacobi_f32", pyamg.ruge_stuben_solver(A3f, max_coarse=10, presmoother=jac, postsmoother=jac), k=5)
    np.random.seed(SEED)
    hier("rs3d_gs_f32", pyamg.ruge_stuben_solver(A3f, max_coarse=10), k=5)
    np.random.seed(SEED)
    E, B = pyamg.gallery.linear_elasticity((14, 14), format="bsr")
    np.random.seed(SEED)
    hier("el2d_blockgs", pyamg.smoothed_aggregation_solver(E, B=B, max_coarse=10))
    np.random.seed(SEED)
    hier("el2d_jacobi", pyamg.smoothed_aggregation_solver(E, B=B, max_coarse=10, presmoother="jacobi",
                                                          postsmoother="jacobi"))
    np.random.seed(SEED)
    hier("el2d_blockjacobi", pyamg.smoothed_aggregation_solver(E, B=B, max_coarse=10, presmoother="block_jacobi",
                                                               postsmoother="block_jacobi"))
    np.random.seed(SEED)
    gs = ("gauss_seidel", {"sweep": "symmetric"})
    hier("el2d_pointgs", pyamg.smoothed_aggregation_solver(E, B=B, max_coarse=10, presmoother=gs, postsmoother=gs))
    # BASELINE config 5 in miniature: 3-D elasticity, BSR(3,3) -> (6,6), P blocks (3,6)/(6,6)
    from tools.problems import elasticity3d
    E3, B3 = elasticity3d(7)
    np.random.seed(SEED)
    hier("el3d_blockgs", pyamg.smoothed_aggregation_solver(E3, B=B3, smooth="jacobi", max_coarse=10), k=6)
    np.random.seed(SEED)
    hier("el3d_blockjacobi", pyamg.smoothed_aggregation_solver(E3, B=B3, smooth="jacobi", max_coarse=10,
                                                               presmoother="block_jacobi",
                                                               postsmoother="block_jacobi"), k=6)
    np.random.seed(SEED)
    hier("el3d_jacobi", pyamg.smoothed_aggregation_solver(E3, B=B3, smooth="jacobi", max_coarse=10,
                                                          presmoother="jacobi", postsmoother="jacobi"), k=6)
    # structurally NON-symmetric operator (first-order upwind convection in x, diffusion in y): the
    # order-exact sweeps then read old values from a snapshot -- inside the captured cycle
    import scipy.sparse as sp
    m = 28
    Dx = sp.diags_array([np.ones(m), -np.ones(m - 1)], offsets=[0, -1], shape=(m, m))
    Dy = sp.diags_array([2 * np.ones(m), -np.ones(m - 1), -np.ones(m - 1)], offsets=[0, -1, 1], shape=(m, m))
    An = sp.csr_array(3.0 * sp.kron(sp.eye_array(m), Dx) + sp.kron(Dy, sp.eye_array(m)))
    An.sort_indices()
    np.random.seed(SEED)
    hier("rs2d_nonsym_gs", pyamg.ruge_stuben_solver(An, max_coarse=10))
    # normal-equation smoothers on the same non-symmetric operator: Kaczmarz (gauss_seidel_ne), Gauss-Seidel on
    # A^H A (gauss_seidel_nr -- what pyamg.solve() configures for non-symmetric matrices, blackbox.py:112-114)
    # and jacobi_ne; plus the blackbox configuration itself (energy-minimisation SA, 2 levels)
    for tag, smo in (("gsnr", ("gauss_seidel_nr", {"sweep": "symmetric", "iterations": 1})),
                     ("gsne", ("gauss_seidel_ne", {"sweep": "symmetric", "iterations": 1})),
                     ("jacobine", ("jacobi_ne", {"iterations": 2}))):
        np.random.seed(SEED)
        hier(f"rs2d_nonsym_{tag}", pyamg.ruge_stuben_solver(An, max_coarse=10, presmoother=smo, postsmoother=smo))
    for tag in ("cgne", "cgnr"):
        np.random.seed(SEED)
        hier(f"rs2d_nonsym_{tag}", pyamg.ruge_stuben_solver(An, max_coarse=10, presmoother=(tag, {"maxiter": 2}),
                                                            postsmoother=(tag, {"maxiter": 2})))
    from pyamg import blackbox
    np.random.seed(SEED)
    hier("bb2d_nonsym_gsnr", blackbox.solver(An, blackbox.solver_configuration(An, verb=False)))
    # AIR (approximate ideal restriction, classical/air.py) on an advection-dominated operator: no
    # presmoother, FC Jacobi (2 F-sweeps, 1 C-sweep of amg_core.jacobi_indexed) as postsmoother, R != P^T
    Aa = sp.csr_array(3.0 * sp.kron(sp.eye_array(m), Dx) + 0.3 * sp.kron(Dy, sp.eye_array(m)))
    Aa.sort_indices()
    np.random.seed(SEED)
    hier("air2d_fcjacobi", pyamg.air_solver(Aa, max_coarse=20))
    # hand-built two-level hierarchy with a CSC restriction (multilevel.py:180-182)
    np.random.seed(SEED)
    ml0 = pyamg.ruge_stuben_solver(A, max_coarse=500, max_levels=2)
    lv = [pyamg.MultilevelSolver.Level(), pyamg.MultilevelSolver.Level()]
    lv[0].A, lv[0].P = ml0.levels[0].A, ml0.levels[0].P          # R omitted -> P.T (csc)
    lv[1].A = ml0.levels[1].A
    mlc = pyamg.MultilevelSolver(lv, coarse_solver="splu")
    from pyamg.relaxation.smoothing import change_smoothers
    change_smoothers(mlc, presmoother=("gauss_seidel", {"sweep": "forward"}),
                     postsmoother=("gauss_seidel", {"sweep": "backward"}))
    hier("rs2d_cscR_splu", mlc)
    # Schwarz smoothers: the reference's doctest configuration (relaxation.py:217-225) and the strength-based variant
    np.random.seed(SEED)
    A20 = pyamg.gallery.poisson((20, 20), format="csr")
    hier("sa2d_schwarz", pyamg.smoothed_aggregation_solver(A20, B=np.ones((A20.shape[0], 1)), coarse_solver="pinv", max_coarse=50,
                                                           presmoother="schwarz", postsmoother=("schwarz", {"sweep": "backward"})), k=5)
    np.random.seed(SEED)
    hier("sa2d_sbschwarz", pyamg.smoothed_aggregation_solver(A20, max_coarse=20, keep=True,
                                                             presmoother=("strength_based_schwarz", {"sweep": "symmetric"}),
                                                             postsmoother=("strength_based_schwarz", {"sweep": "symmetric"})), k=4)
    # CF / FC block Jacobi (relaxation.py:1271-1411, amg_core.block_jacobi_indexed) on the 2-D elasticity hierarchy:
    # the registry needs a C/F splitting per level (an AIR-style solver on a block system would bring its own; here a
    # seeded one is attached to the SA levels), the smoothers are then the reference's own bound callables
    np.random.seed(SEED)
    mlb = pyamg.smoothed_aggregation_solver(E, B=B, max_coarse=10)
    rs = np.random.RandomState(SEED + 21)
    for lvl in mlb.levels[:-1]:
        lvl.splitting = rs.rand(lvl.A.shape[0] // lvl.A.blocksize[0]) < 0.35
    change_smoothers(mlb, presmoother=("cf_block_jacobi", {"omega": 0.8, "f_iterations": 2, "c_iterations": 1}),
                     postsmoother=("fc_block_jacobi", {"omega": 0.9, "iterations": 2}))
    hier("el2d_cfblockjacobi", mlb)


def make_kernels():
    rng = np.random.RandomState(SEED + 1)
    out = {}
    A = pyamg.gallery.poisson((23, 17), format="csr")
    n = A.shape[0]
    # irregular, non-symmetric pattern with an empty row, a missing diagonal and a zero diagonal
    import scipy.sparse as sp
    G = sp.random(300, 300, density=0.03, random_state=rng, format="lil")
    G.setdiag(rng.rand(300) + 1.0)
    G[5, :] = 0
    G[9, 9] = 0.0
    G = sp.csr_array(G.tocsr())
    G.data[G.indptr[9]:G.indptr[10]][G.indices[G.indptr[9]:G.indptr[10]] == 9] = 0.0   # explicit zero diagonal
    G.sort_indices()
    for tag, M in (("pois", A), ("irr", G)):
        m = M.shape[0]
        x = rng.rand(m)
        b = rng.rand(m)
        out[f"{tag}.indptr"], out[f"{tag}.indices"], out[f"{tag}.data"] = M.indptr, M.indices, M.data
        out[f"{tag}.x"], out[f"{tag}.b"] = x, b
        out[f"{tag}.Ax"] = M @ x
        for sweep in ("forward", "backward", "symmetric"):
            y = x.copy()
            rr.gauss_seidel(M, y, b, iterations=2, sweep=sweep)
            out[f"{tag}.gs.{sweep}"] = y
            y = x.copy()
            rr.sor(M, y, b, 1.3, iterations=2, sweep=sweep)
            out[f"{tag}.sor.{sweep}"] = y
        y = x.copy()
        rr.jacobi(M, y, b, iterations=3, omega=0.8)
        out[f"{tag}.jacobi"] = y
        y = x.copy()
        rr.polynomial(M, y, b, coefficients=np.array([0.05, -0.4, 0.9]), iterations=2)
        out[f"{tag}.poly"] = y
        y = np.zeros(m)
        rr.polynomial(M, y, b, coefficients=np.array([0.05, -0.4, 0.9]), iterations=1)
        out[f"{tag}.poly0"] = y
        # BSR(1,1) flavour (what SA levels >= 1 look like)
        Mb = M.tobsr(blocksize=(1, 1))
        y = x.copy()
        rr.jacobi(Mb, y, b, iterations=2, omega=0.8)
        out[f"{tag}.bsr1.jacobi"] = y
        y = x.copy()
        rr.gauss_seidel(Mb, y, b, iterations=1, sweep="symmetric")
        out[f"{tag}.bsr1.gs"] = y
    # block matrices
    E, _ = pyamg.gallery.linear_elasticity((9, 9), format="bsr")
    E = E.tobsr(blocksize=(2, 2))
    m = E.shape[0]
    x = rng.rand(m)
    b = rng.rand(m)
    from pyamg.util.utils import get_block_diag
    out["el.indptr"], out["el.indices"], out["el.data"] = E.indptr, E.indices, E.data.reshape(-1)
    out["el.x"], out["el.b"] = x, b
    out["el.Ax"] = E @ x
    Dinv = get_block_diag(E, blocksize=2, inv_flag=True)
    out["el.Dinv"] = Dinv
    y = x.copy(); rr.jacobi(E, y, b, iterations=2, omega=0.6); out["el.jacobi"] = y
    y = x.copy(); rr.gauss_seidel(E, y, b, iterations=1, sweep="symmetric"); out["el.gs"] = y
    y = x.copy(); rr.block_jacobi(E, y, b, Dinv=Dinv, blocksize=2, iterations=2, omega=0.7); out["el.bjacobi"] = y
    y = x.copy(); rr.block_gauss_seidel(E, y, b, iterations=1, sweep="symmetric", blocksize=2, Dinv=Dinv); out["el.bgs"] = y
    # non-square blocks (P of an elasticity SA hierarchy): BSR (2,3)
    np.random.seed(SEED)
    Ef, B = pyamg.gallery.linear_elasticity((9, 9), format="bsr")
    ml = pyamg.smoothed_aggregation_solver(Ef, B=B, max_coarse=10)
    P = ml.levels[0].P
    xc = rng.rand(P.shape[1])
    out["P.indptr"], out["P.indices"], out["P.data"] = P.indptr, P.indices, P.data.reshape(-1)
    out["P.meta"] = np.array([P.shape[0], P.shape[1], P.blocksize[0], P.blocksize[1]])
    out["P.x"] = xc
    out["P.Ax"] = P @ xc
    np.savez_compressed(HERE / "kernels.npz", **out)
    print("kernels.npz:", len(out), "arrays")


def make_known_answers():
    """Known answers pinned by the reference's own tests / doctests / fixture."""
    ka = {
        "_source": "restated from the reference's tests; each entry cites file:line",
        # A = tridiag(-1, 2, -1) of size N (the tests build it with diags_array)
        "jacobi": {"cite": "pyamg/relaxation/tests/test_relaxation.py:148-197", "cases": [
            {"N": 1, "x": [0.0], "b": [0.0], "omega": 1.0, "expect": [0.0]},
            {"N": 3, "x": [0.0, 0.0, 0.0], "b": [0.0, 1.0, 2.0], "omega": 1.0, "expect": [0.0, 0.5, 1.0]},
            {"N": 3, "x": [0.0, 1.0, 2.0], "b": [0.0, 0.0, 0.0], "omega": 1.0, "expect": [0.5, 1.0, 0.5]},
            {"N": 1, "x": [0.0], "b": [10.0], "omega": 1.0, "expect": [5.0]},
            {"N": 3, "x": [0.0, 1.0, 2.0], "b": [10.0, 20.0, 30.0], "omega": 1.0, "expect": [5.5, 11.0, 15.5]},
            {"N": 3, "x": [0.0, 1.0, 2.0], "b": [10.0, 20.0, 30.0], "omega": 1.0 / 3.0,
             "expect": [2.0 / 3.0 * 0.0 + 5.5 / 3.0, 2.0 / 3.0 * 1.0 + 11.0 / 3.0, 2.0 / 3.0 * 2.0 + 15.5 / 3.0]}]},
        "gauss_seidel": {"cite": "pyamg/relaxation/tests/test_relaxation.py:299-346", "cases": [
            {"N": 1, "x": [0.0], "b": [0.0], "sweep": "forward", "expect": [0.0]},
            {"N": 3, "x": [0.0, 1.0, 2.0], "b": [0.0, 0.0, 0.0], "sweep": "forward",
             "expect": [1.0 / 2.0, 5.0 / 4.0, 5.0 / 8.0]},
            {"N": 1, "x": [0.0], "b": [0.0], "sweep": "backward", "expect": [0.0]},
            {"N": 3, "x": [0.0, 1.0, 2.0], "b": [0.0, 0.0, 0.0], "sweep": "backward",
             "expect": [1.0 / 8.0, 1.0 / 4.0, 1.0 / 2.0]},
            {"N": 1, "x": [0.0], "b": [10.0], "sweep": "forward", "expect": [5.0]},
            {"N": 3, "x": [0.0, 1.0, 2.0], "b": [10.0, 20.0, 30.0], "sweep": "forward",
             "expect": [11.0 / 2.0, 55.0 / 4.0, 175.0 / 8.0]}]},
        "gauss_seidel_200": {"cite": "pyamg/relaxation/tests/test_relaxation.py:348-362", "N": 100,
                             "iterations": 200, "resid_below": 0.01},
        "sor_wikipedia": {"cite": "pyamg/relaxation/tests/test_relaxation.py:808-835",
                          "A": [[4.0, -1.0, -6.0, 0.0], [-5.0, -4.0, 10.0, 8.0], [0.0, 9.0, 4.0, -2.0],
                                [1.0, 0.0, -7.0, 5.0]],
                          "b": [2.0, 21.0, -12.0, -6.0], "omega": 0.5,
                          "expect_after": {"1": [0.25, -2.78125, 1.6289062, 0.5152344],
                                           "2": [1.2490234, -2.2448974, 1.9687712, 0.9108547],
                                           "3": [2.070478, -1.6696789, 1.5904881, 0.76172125],
                                           "38": [3.0, -2.0, 2.0, 1.0]}, "rtol": 1e-6},
        "doctest_sor_norm": {"cite": "pyamg/relaxation/relaxation.py:130-138", "A": "poisson((10,10))",
                             "omega": 1.33, "iterations": 10, "expect_norm_3dec": 2.013},
        "doctest_gs_norm": {"cite": "pyamg/relaxation/relaxation.py:291-299", "A": "poisson((10,10))",
                            "iterations": 10, "expect_norm_3dec": 4.007},
        "doctest_jacobi_norm": {"cite": "pyamg/relaxation/relaxation.py:373-381", "A": "poisson((10,10))",
                                "iterations": 10, "omega": 1.0, "expect_norm_3dec": 5.835},
        "paper_example": {"cite": "docs/paper/example.res.txt:15-36 (generator docs/paper/example.py:5-18)",
                          "residuals": [float(v) for v in
                                        np.loadtxt("/root/reference/docs/paper/example.res.txt")]},
    }
    # verify the restated doctest values against the reference before writing them down
    A = pyamg.gallery.poisson((10, 10), format="csr")
    b = np.ones((A.shape[0], 1))
    from pyamg.util.linalg import norm
    x0 = np.zeros((A.shape[0], 1)); rr.sor(A, x0, b, 1.33, iterations=10)
    assert f"{norm(b - A @ x0):2.4}" == "2.013"
    x0 = np.zeros((A.shape[0], 1)); rr.gauss_seidel(A, x0, b, iterations=10)
    assert f"{norm(b - A @ x0):2.4}" == "4.007"
    x0 = np.zeros((A.shape[0], 1)); rr.jacobi(A, x0, b, iterations=10, omega=1.0)
    assert f"{norm(b - A @ x0):2.4}" == "5.835"
    (HERE / "known_answers.json").write_text(json.dumps(ka, indent=1))
    print("known_answers.json written; sor doctest norm =", ka["doctest_sor_norm"]["expect_norm_3dec"])


def make_kernels_indexed():
    """amg_core.jacobi_indexed and relaxation.cf_jacobi / fc_jacobi of the reference -> kernels_indexed.npz"""
    import scipy.sparse as sp
    from pyamg import amg_core
    rng = np.random.RandomState(SEED + 11)
    out = {}
    G = sp.random(400, 400, density=0.03, random_state=rng, format="lil")
    G.setdiag(rng.rand(400) + 1.0)
    G[7, :] = 0                       # empty row
    G[11, 11] = 0.0                   # missing diagonal
    G = sp.csr_array(G.tocsr())
    G.sort_indices()
    P = pyamg.gallery.poisson((21, 19), format="csr")
    for tag, M in (("irr", G), ("pois", P)):
        for dt in (np.float64, np.float32):
            Md = sp.csr_array(M.astype(dt))
            n = Md.shape[0]
            x = rng.rand(n).astype(dt); b = rng.rand(n).astype(dt)
            idx = rng.permutation(n)[: n // 3].astype(np.int32)
            F = np.sort(rng.permutation(n)[: (2 * n) // 3]).astype(np.int32)
            Cp = np.setdiff1d(np.arange(n, dtype=np.int32), F).astype(np.int32)
            k = f"{tag}_{np.dtype(dt).name}"
            out[f"{k}.indptr"], out[f"{k}.indices"], out[f"{k}.data"] = Md.indptr, Md.indices, Md.data
            out[f"{k}.x"], out[f"{k}.b"], out[f"{k}.idx"], out[f"{k}.F"], out[f"{k}.C"] = x, b, idx, F, Cp
            y = x.copy(); amg_core.jacobi_indexed(Md.indptr, Md.indices, Md.data, y, b, idx, np.array([0.7], dtype=dt))
            out[f"{k}.jacobi_indexed"] = y
            y = x.copy(); rr.fc_jacobi(Md, y, b, Cp, F, iterations=2, f_iterations=2, c_iterations=1, omega=0.9)
            out[f"{k}.fc_jacobi"] = y
            y = x.copy(); rr.cf_jacobi(Md, y, b, Cp, F, iterations=1, f_iterations=1, c_iterations=2, omega=1.0)
            out[f"{k}.cf_jacobi"] = y
    np.savez_compressed(HERE / "kernels_indexed.npz", **out)
    print("kernels_indexed.npz written:", len(out), "arrays")


def make_kernels_schwarz():
    """amg_core.overlapping_schwarz_csr / relaxation.schwarz of the reference -> kernels_schwarz.npz (the subdomains and
    inverted blocks are the reference's own schwarz_parameters)."""
    import scipy.sparse as sp
    from pyamg import amg_core
    rng = np.random.RandomState(SEED + 23)
    out = {}
    G = sp.random(260, 260, density=0.03, random_state=rng, format="csr")
    G = sp.csr_array(G + G.T + 6.0 * sp.eye_array(260))
    G.sort_indices()
    P = pyamg.gallery.poisson((16, 14), format="csr")
    for tag, M in (("irr", G), ("pois", P)):
        for dt in (np.float64, np.float32):
            Md = sp.csr_array(M.astype(dt))
            Md.sort_indices()
            n = Md.shape[0]
            x = rng.rand(n).astype(dt); b = rng.rand(n).astype(dt)
            sub, sptr, inv, iptr = rr.schwarz_parameters(Md)
            k = f"{tag}_{np.dtype(dt).name}"
            out[f"{k}.indptr"], out[f"{k}.indices"], out[f"{k}.data"] = Md.indptr, Md.indices, Md.data
            out[f"{k}.x"], out[f"{k}.b"] = x, b
            out[f"{k}.sub"], out[f"{k}.sptr"], out[f"{k}.inv"], out[f"{k}.iptr"] = sub, sptr, inv, iptr
            for sweep in ("forward", "backward", "symmetric"):
                y = x.copy(); rr.schwarz(Md, y, b, iterations=2, sweep=sweep)
                out[f"{k}.{sweep}"] = y
            y = x.copy()
            amg_core.overlapping_schwarz_csr(Md.indptr, Md.indices, Md.data, y, b, inv, iptr, sub, sptr, len(sptr) - 1, n, 3, n - 1, 2)
            out[f"{k}.strided"] = y
        if tag == "irr":
            continue
        # fewer, larger subdomains (two-hop neighbourhoods of every third row)
        Md = sp.csr_array(M.astype(np.float64)); Md.sort_indices()
        C2 = sp.csr_array(Md @ Md); C2.sort_indices()
        rows = np.arange(0, Md.shape[0], 3)
        sptr = np.zeros(len(rows) + 1, dtype=np.int32)
        sptr[1:] = np.cumsum([C2.indptr[r + 1] - C2.indptr[r] for r in rows])
        sub = np.concatenate([C2.indices[C2.indptr[r]:C2.indptr[r + 1]] for r in rows]).astype(np.int32)
        x = rng.rand(Md.shape[0]); b = rng.rand(Md.shape[0])
        Mc = sp.csr_array(Md.copy())
        y = x.copy(); rr.schwarz(Mc, y, b, iterations=1, subdomain=sub, subdomain_ptr=sptr, sweep="symmetric")
        k = f"{tag}_big"
        out[f"{k}.indptr"], out[f"{k}.indices"], out[f"{k}.data"] = Md.indptr, Md.indices, Md.data
        out[f"{k}.x"], out[f"{k}.b"], out[f"{k}.sub"], out[f"{k}.sptr"] = x, b, sub, sptr
        out[f"{k}.inv"], out[f"{k}.iptr"] = Mc.schwarz_parameters[2], Mc.schwarz_parameters[3]
        out[f"{k}.symmetric"] = y
    np.savez_compressed(HERE / "kernels_schwarz.npz", **out)
    print("kernels_schwarz.npz written:", len(out), "arrays")


def make_kernels_gsidx():
    """amg_core.gauss_seidel_indexed / relaxation.gauss_seidel_indexed of the reference -> kernels_gsidx.npz"""
    import scipy.sparse as sp
    from pyamg import amg_core
    rng = np.random.RandomState(SEED + 19)
    out = {}
    G = sp.random(300, 300, density=0.04, random_state=rng, format="lil")
    G.setdiag(rng.rand(300) + 1.0)
    G[9, :] = 0                       # empty row
    G[17, 17] = 0.0                   # missing diagonal
    G = sp.csr_array(G.tocsr())
    G.sort_indices()
    P = pyamg.gallery.poisson((17, 15), format="csr")
    for tag, M in (("irr", G), ("pois", P)):
        for dt in (np.float64, np.float32):
            Md = sp.csr_array(M.astype(dt))
            n = Md.shape[0]
            x = rng.rand(n).astype(dt); b = rng.rand(n).astype(dt)
            idx = rng.permutation(n)[: (2 * n) // 3].astype(np.int32)           # no row twice, arbitrary order
            dup = np.concatenate([idx[:40], idx[20:60][::-1], idx[:10]]).astype(np.int32)    # rows listed two and three times
            k = f"{tag}_{np.dtype(dt).name}"
            out[f"{k}.indptr"], out[f"{k}.indices"], out[f"{k}.data"] = Md.indptr, Md.indices, Md.data
            out[f"{k}.x"], out[f"{k}.b"], out[f"{k}.idx"], out[f"{k}.dup"] = x, b, idx, dup
            for sweep in ("forward", "backward", "symmetric"):
                y = x.copy(); rr.gauss_seidel_indexed(Md, y, b, idx, iterations=2, sweep=sweep)
                out[f"{k}.{sweep}"] = y
            y = x.copy(); rr.gauss_seidel_indexed(Md, y, b, dup, iterations=1, sweep="forward")
            out[f"{k}.dup.forward"] = y
            y = x.copy(); amg_core.gauss_seidel_indexed(Md.indptr, Md.indices, Md.data, y, b, idx, 1, len(idx) - 1, 2)   # strided slice of the list
            out[f"{k}.strided"] = y
    np.savez_compressed(HERE / "kernels_gsidx.npz", **out)
    print("kernels_gsidx.npz written:", len(out), "arrays")


def make_kernels_blockidx():
    """amg_core.block_jacobi_indexed and relaxation.cf_block_jacobi / fc_block_jacobi of the reference -> kernels_blockidx.npz"""
    import scipy.sparse as sp
    from pyamg import amg_core
    from pyamg.util.utils import get_block_diag
    rng = np.random.RandomState(SEED + 17)
    out = {}
    for tag, bs, nb in (("b3", 3, 90), ("b2", 2, 150)):
        G = sp.random(nb, nb, density=0.06, random_state=rng, format="lil")
        G.setdiag(1.0)
        G[5, :] = 0                                  # empty block row (no diagonal block either)
        pat = sp.csr_array(G.tocsr())
        pat.eliminate_zeros()
        blocks = rng.rand(pat.nnz, bs, bs) - 0.3
        M = sp.bsr_array((blocks, pat.indices, pat.indptr), shape=(nb * bs, nb * bs))
        for i in range(nb):                          # dominant diagonal blocks
            for p in range(M.indptr[i], M.indptr[i + 1]):
                if M.indices[p] == i:
                    M.data[p] += 4.0 * np.eye(bs)
        for dt in (np.float64, np.float32):
            Md = sp.bsr_array((M.data.astype(dt), M.indices, M.indptr), shape=M.shape)
            n = Md.shape[0]
            x = rng.rand(n).astype(dt); b = rng.rand(n).astype(dt)
            Dinv = get_block_diag(Md, blocksize=bs, inv_flag=True).astype(dt)
            idx = rng.permutation(nb)[: nb // 3].astype(np.int32)
            F = np.sort(rng.permutation(nb)[: (2 * nb) // 3]).astype(np.int32)
            Cp = np.setdiff1d(np.arange(nb, dtype=np.int32), F).astype(np.int32)
            k = f"{tag}_{np.dtype(dt).name}"
            out[f"{k}.indptr"], out[f"{k}.indices"], out[f"{k}.data"] = Md.indptr, Md.indices, Md.data
            out[f"{k}.x"], out[f"{k}.b"], out[f"{k}.idx"], out[f"{k}.F"], out[f"{k}.C"], out[f"{k}.Dinv"] = x, b, idx, F, Cp, Dinv
            y = x.copy()
            amg_core.block_jacobi_indexed(Md.indptr, Md.indices, np.ravel(Md.data), y, b, np.ravel(Dinv), idx, np.array([0.7], dtype=dt), bs)
            out[f"{k}.block_jacobi_indexed"] = y
            y = x.copy(); rr.fc_block_jacobi(Md, y, b, Cp, F, Dinv=Dinv, blocksize=bs, iterations=2, f_iterations=2, c_iterations=1, omega=0.9)
            out[f"{k}.fc_block_jacobi"] = y
            y = x.copy(); rr.cf_block_jacobi(Md, y, b, Cp, F, Dinv=Dinv, blocksize=bs, iterations=1, f_iterations=1, c_iterations=2, omega=1.0)
            out[f"{k}.cf_block_jacobi"] = y
    np.savez_compressed(HERE / "kernels_blockidx.npz", **out)
    print("kernels_blockidx.npz written:", len(out), "arrays")


def make_kernels_ne():
    """relaxation.gauss_seidel_ne / gauss_seidel_nr / jacobi_ne of the reference -> kernels_ne.npz (float64: the
    reference hands a float64 Dinv to its float32 kernels, i.e. raises TypeError there)."""
    import scipy.sparse as sp
    rng = np.random.RandomState(SEED + 13)
    out = {}
    G = sp.random(350, 350, density=0.03, random_state=rng, format="lil")
    G.setdiag(rng.rand(350) + 1.0)
    G[7, :] = 0                       # empty row
    G[:, 13] = 0                      # empty column
    G = sp.csr_array(G.tocsr())
    G.eliminate_zeros()
    G.sort_indices()
    P = sp.csr_array(pyamg.gallery.poisson((19, 17), format="csr"))
    for tag, M in (("irr", G), ("pois", P)):
        n = M.shape[0]
        x = rng.rand(n); b = rng.rand(n)
        out[f"{tag}.indptr"], out[f"{tag}.indices"], out[f"{tag}.data"] = M.indptr, M.indices, M.data
        out[f"{tag}.x"], out[f"{tag}.b"] = x, b
        for sweep in ("forward", "backward", "symmetric"):
            y = x.copy(); rr.gauss_seidel_ne(M, y, b, iterations=2, sweep=sweep, omega=0.9)
            out[f"{tag}.gauss_seidel_ne.{sweep}"] = y
            y = x.copy(); rr.gauss_seidel_nr(sp.csc_array(M), y, b, iterations=2, sweep=sweep, omega=1.1)
            out[f"{tag}.gauss_seidel_nr.{sweep}"] = y
        y = x.copy(); rr.jacobi_ne(M, y, b, iterations=3, omega=0.6)
        out[f"{tag}.jacobi_ne"] = y
    np.savez_compressed(HERE / "kernels_ne.npz", **out)
    print("kernels_ne.npz written:", len(out), "arrays")


def make_kernels_setup():
    """setup-phase kernels: amg_core.pinv_array on seeded blocks (regular, singular, rank-deficient,
    zero, tiny), both storage conventions, f64 and f32; get_block_diag of an elasticity operator"""
    from pyamg import amg_core
    from pyamg.util.utils import get_block_diag
    rng = np.random.default_rng(SEED)
    out = {}
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        for n in (1, 2, 3, 4, 5, 6):
            m = 200
            A = rng.standard_normal((m, n, n)).astype(dt)
            A[0] = 0
            A[4] = np.eye(n) * 1e-30
            A[5] = np.eye(n) * 3.0
            if n > 1:
                A[1, :, 0] = A[1, :, 1]
                A[2] = np.ones((n, n))
                A[3] = np.diag(np.arange(n)).astype(dt)
                A[6] = np.triu(A[6])
            out[f"pinv.{tag}.{n}.in"] = A.copy()
            for tr in ("T", "F"):
                R = A.copy()
                amg_core.pinv_array(R.ravel(), m, n, tr)
                out[f"pinv.{tag}.{n}.{tr}"] = R
    A, _ = pyamg.gallery.linear_elasticity((12, 12), format="bsr")
    out["bd.indptr"], out["bd.indices"], out["bd.data"] = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    out["bd.shape"] = np.array(A.shape)
    out["bd.inv2"] = get_block_diag(A.copy(), blocksize=2, inv_flag=True)
    out["bd.blk2"] = get_block_diag(A.copy(), blocksize=2, inv_flag=False)
    out["bd.inv4"] = get_block_diag(pyamg.gallery.poisson((10, 10), format="csr"), blocksize=4, inv_flag=True)
    # amg_core.standard_aggregation on strength graphs (2-D / 3-D stencils, an irregular symmetric pattern, isolated nodes)
    # and amg_core.fit_candidates / tentative.fit_candidates on the aggregates (scalar, vector and rank-deficient candidates)
    import scipy.sparse as sp
    from pyamg.aggregation.aggregate import standard_aggregation
    from pyamg.aggregation.tentative import fit_candidates
    from pyamg.strength import symmetric_strength_of_connection
    graphs = {"p2d": symmetric_strength_of_connection(pyamg.gallery.poisson((40, 40), format="csr"), theta=0.0),
              "p3d": symmetric_strength_of_connection(pyamg.gallery.poisson((14, 14, 14), format="csr"), theta=0.0)}
    M = sp.random_array((600, 600), density=0.002, random_state=rng, format="csr")
    M = (M + M.T).tocsr()
    M.setdiag(0)
    M.eliminate_zeros()                                      # some nodes end up without any neighbour
    M = (M + sp.diags_array((np.arange(600) % 3 == 0).astype(float))).tocsr()     # diagonal entries on a third of the rows
    M.eliminate_zeros()
    graphs["irr"] = M
    aniso = pyamg.gallery.stencil_grid(np.array([[0, -1.0, 0], [-0.001, 2.002, -0.001], [0, -1.0, 0]]), (30, 30), format="csr")
    graphs["aniso"] = symmetric_strength_of_connection(aniso, theta=0.25)
    for name, Cg in graphs.items():
        Cg = sp.csr_array(Cg)
        n = Cg.shape[0]
        x, y = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        cnt = amg_core.standard_aggregation(n, Cg.indptr.astype(np.int32), Cg.indices.astype(np.int32), x, y)
        out[f"agg.{name}.indptr"], out[f"agg.{name}.indices"] = Cg.indptr.astype(np.int32), Cg.indices.astype(np.int32)
        out[f"agg.{name}.x"], out[f"agg.{name}.y"] = x, y[:cnt]
    AggOp, _ = standard_aggregation(sp.csr_array(graphs["p2d"]))
    out["fit.Tp"], out["fit.Tj"] = AggOp.indptr.astype(np.int32), AggOp.indices.astype(np.int32)
    out["fit.shape"] = np.array(AggOp.shape)
    for tag, K1, K2, dt in (("a", 1, 1, np.float64), ("b", 2, 3, np.float64), ("c", 3, 6, np.float64), ("d", 1, 2, np.float32)):
        B = rng.standard_normal((AggOp.shape[0] * K1, K2)).astype(dt)
        if K2 > 1:
            B[:, 1] = 2 * B[:, 0]                            # a dependent candidate: its column of Q is zeroed
        Q, R = fit_candidates(AggOp, B)
        out[f"fit.{tag}.B"], out[f"fit.{tag}.Q"], out[f"fit.{tag}.R"] = B, Q.data, R
        assert np.array_equal(Q.indptr, AggOp.indptr) and np.array_equal(Q.indices, AggOp.indices)
    np.savez_compressed(HERE / "kernels_setup.npz", **out)
    print("kernels_setup.npz written:", len(out), "arrays")


def save_accel():
    if ACCEL:
        np.savez_compressed(HERE / "accel_fgmres.npz", **ACCEL)
        print("accel_fgmres.npz written:", len(ACCEL), "arrays")
    if ACCEL_CG:
        np.savez_compressed(HERE / "accel_cg.npz", **ACCEL_CG)
        print("accel_cg.npz written:", len(ACCEL_CG), "arrays")


if __name__ == "__main__" and "--indexed-only" in sys.argv:
    make_kernels_indexed()
    sys.exit(0)

if __name__ == "__main__" and "--schwarz-only" in sys.argv:
    make_kernels_schwarz()
    sys.exit(0)

if __name__ == "__main__" and "--gsidx-only" in sys.argv:
    make_kernels_gsidx()
    sys.exit(0)

if __name__ == "__main__" and "--blockidx-only" in sys.argv:
    make_kernels_blockidx()
    sys.exit(0)

if __name__ == "__main__" and "--setup-only" in sys.argv:
    make_kernels_setup()
    sys.exit(0)

if __name__ == "__main__" and "--ne-only" in sys.argv:
    make_kernels_ne()
    sys.exit(0)

if __name__ == "__main__" and (ONLY or ACCEL_ONLY):
    make_hierarchies()
    save_accel()
    sys.exit(0)

if __name__ == "__main__":
    if "--hier-only" not in sys.argv:
        make_known_answers()
        make_kernels()
        make_kernels_indexed()
        make_kernels_blockidx()
        make_kernels_gsidx()
        make_kernels_schwarz()
        make_kernels_ne()
        make_kernels_setup()
    make_hierarchies()
    save_accel()
