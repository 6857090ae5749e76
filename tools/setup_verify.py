#!/usr/bin/env python3
"""Diagnostic: build an SA hierarchy under device_setup with every routed operation ALSO computed by the reference /
SciPy and compared (products and differences array for array, strength array for array, the smoothed prolongator
against SciPy's expression with the device's spectral radius); prints one line per operation."""
import argparse, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401
import pyamg
import scipy.sparse as sp
import pyamg_amd.aggregation as ag

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[128, 128, 128])
a = ap.parse_args()
orig_mm = {cls: cls._matmul_sparse for cls in (sp.csr_array, sp.bsr_array, sp.csr_matrix, sp.bsr_matrix)}


def same(C, ref):
    return (C.format == ref.format and C.nnz == ref.nnz and np.array_equal(C.indptr, ref.indptr)
            and np.array_equal(C.indices, ref.indices) and np.array_equal(np.ravel(C.data), np.ravel(ref.data)))


def detail(C, ref):
    if C.nnz != ref.nnz or not np.array_equal(C.indptr, ref.indptr):
        d = np.flatnonzero(np.diff(C.indptr) != np.diff(ref.indptr))
        return f"row lengths differ in {d.size} rows, first {d[:3]}"
    bad = np.flatnonzero((C.indices != ref.indices) | (np.ravel(C.data) != np.ravel(ref.data)))
    return f"{bad.size} entries differ, first at {bad[:3]} (of {C.nnz}), max |d| {np.abs(np.ravel(C.data)[bad] - np.ravel(ref.data)[bad]).max() if bad.size else 0}"


dev_product = ag._device_product


def checked_product(self, other):
    t = time.time(); C = dev_product(self, other); td = time.time() - t
    t = time.time(); ref = orig_mm[type(self)](self, other); tr = time.time() - t
    ok = same(C, ref)
    print(f"product {self.format}{self.shape} @ {other.format}{other.shape}: nnz {C.nnz} dev {td:.2f}s scipy {tr:.2f}s {'same' if ok else 'MISMATCH ' + detail(C, ref)}", flush=True)
    return C


ref_soc = pyamg.strength.symmetric_strength_of_connection
dev_soc = ag.symmetric_strength_of_connection


def checked_soc(A, theta=0):
    S = dev_soc(A, theta)
    ref = ref_soc(A, theta)
    print(f"strength {A.format}{A.shape}: nnz {S.nnz} {'same' if same(S, ref) else 'MISMATCH ' + detail(S, ref)}", flush=True)
    return S


dev_rho = ag._spectral_radius
last = {}


def checked_rho(dm, tol, maxiter, restart, v0, want_vector=False):
    out = dev_rho(dm, tol, maxiter, restart, v0, want_vector)
    last["rho"] = out[0] if want_vector else out
    print(f"spectral radius n={dm.shape[0]}: {last['rho']!r}", flush=True)
    return out


dev_jac = ag.jacobi_prolongation_smoother


def checked_jac(S, T, C, B, omega=4.0 / 3.0, degree=1, filter_entries=False, weighting="diagonal"):
    P = dev_jac(S, T, C, B, omega=omega, degree=degree, filter_entries=filter_entries, weighting=weighting)
    rho = last["rho"]
    D = S.diagonal()
    Dinv = np.zeros_like(D); Dinv[D != 0] = 1.0 / D[D != 0]
    W = S.copy()
    W.data = np.ravel(W.data) * np.repeat(Dinv, np.diff(W.indptr))        # scale_rows (scalar storage)
    if W.format == "bsr":
        W.data = W.data.reshape(-1, 1, 1)
    W = (omega / rho) * W
    U = orig_mm[type(W)](W, T)
    ref = T - U
    print(f"prolongation smoother S{S.shape} T{T.shape}: nnz {P.nnz} {'same' if same(P, ref) else 'MISMATCH ' + detail(P, ref)}", flush=True)
    return P


ag._device_product = checked_product
ag.symmetric_strength_of_connection = checked_soc
ag._spectral_radius = checked_rho
ag.jacobi_prolongation_smoother = checked_jac
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
np.random.seed(1)
t = time.time()
with ag.device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
print(f"setup with checks {time.time() - t:.1f}s; levels {[(L.A.shape[0], L.A.nnz) for L in ml.levels]}", flush=True)
b = np.zeros(A.shape[0]); x0 = np.random.rand(A.shape[0])
from pyamg_amd import DeviceMultilevelSolver
res = []
DeviceMultilevelSolver(ml).solve(b, x0=x0, tol=1e-30, maxiter=4, residuals=res)
print("residuals of 4 device cycles (b = 0):", [f"{r:.4e}" for r in res], flush=True)
