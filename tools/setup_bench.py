#!/usr/bin/env python3
"""Setup-phase timing: the reference's smoothed_aggregation_solver (oracle/_ref, serial host code) as it is, and with the
device setup operators patched in (pyamg_amd.aggregation.device_setup); the Galerkin products level by level, SciPy's
vs the device's.  Prints one JSON line."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401  (test / bench infrastructure: the reference builds the hierarchy)
import pyamg
import scipy.sparse as sp
from pyamg_amd.aggregation import device_setup, galerkin_product

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[128, 128, 128])
ap.add_argument("--rap", type=int, default=1)
a = ap.parse_args()
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
out = {"grid": a.grid, "n": A.shape[0]}
galerkin_product(sp.csr_array(np.eye(4)), sp.csr_array(np.eye(4)), sp.csr_array(np.eye(4)))      # context + kernel load
np.random.seed(1)
t = time.time()
ml_ref = pyamg.smoothed_aggregation_solver(A.copy(), max_coarse=10)
out["reference_setup_s"] = round(time.time() - t, 2)
np.random.seed(1)
t = time.time()
with device_setup(pyamg):
    ml_dev = pyamg.smoothed_aggregation_solver(A.copy(), max_coarse=10)
out["device_setup_s"] = round(time.time() - t, 2)
out["levels"] = [[int(L.A.shape[0]), int(L.A.nnz)] for L in ml_dev.levels]
out["levels_match"] = [[int(L.A.shape[0]), int(L.A.nnz)] for L in ml_ref.levels] == out["levels"]
out["max_rel_diff_level_A"] = [float(abs(sp.csr_array(Ld.A) - sp.csr_array(Lr.A)).max() / abs(Lr.A).max()) for Ld, Lr in zip(ml_dev.levels, ml_ref.levels)]
if a.rap:
    rap = []
    for Lr, Ln in zip(ml_ref.levels[:-1], ml_ref.levels[1:]):
        t = time.time(); ref = Lr.R @ Lr.A @ Lr.P; t_ref = time.time() - t
        t = time.time(); Ac = galerkin_product(Lr.R, Lr.A, Lr.P); t_dev = time.time() - t
        rap.append({"n": int(Lr.A.shape[0]), "scipy_s": round(t_ref, 3), "device_s": round(t_dev, 3),
                    "same_arrays": bool(Ac.format == ref.format and np.array_equal(Ac.indptr, ref.indptr) and np.array_equal(Ac.indices, ref.indices)
                                        and np.array_equal(np.ravel(Ac.data), np.ravel(ref.data)))})
    out["galerkin"] = rap
print(json.dumps(out))
