"""pyamg_amd -- MI355X-native AMG solve-phase engine.

Drop-in for the solve phase behind ``pyamg.multilevel.MultilevelSolver.solve()``:
hand-written HIP kernels for gfx950 (CSR/BSR SpMV, weighted Jacobi, order-exact
Gauss-Seidel/SOR, polynomial/Chebyshev, block relaxation) behind a plain C ABI
(``include/pyamg_amd.h``); the hierarchy setup stays in the reference on the host.

    ml  = pyamg.smoothed_aggregation_solver(A)          # reference, host
    dml = pyamg_amd.DeviceMultilevelSolver(ml)          # ship levels to HBM once
    x   = dml.solve(b, tol=1e-8)                        # same signature / semantics

Importing the package never touches the GPU; using it without the built HIP library or
without a device raises (no CPU fallback).
"""
from .hierarchy import HierarchySpec, LevelSpec, SmootherSpec, SparseOp, extract  # noqa: F401
from .multilevel import DeviceMatrix, DeviceMultilevelSolver  # noqa: F401
from . import amg_core, relaxation  # noqa: F401

__version__ = "0.1.0"
