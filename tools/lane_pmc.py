#!/usr/bin/env python3
"""Forward fast-order Gauss-Seidel sweeps on levels 0..2 of the 256^3 SA hierarchy, nothing else: the command
tools/pmc_lane_probe.py runs under rocprofv3 --pmc (memory requests, stalls, latencies of gs_lane / gs_line).  Not product code."""
import sys, json
from pathlib import Path
import numpy as np
import scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix

levels = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1+2").split("+")]
tune = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
cache = Path("/tmp/pamg_levels_256.npz")
if not cache.exists():
    import oracle.refimport  # noqa
    import pyamg
    from pyamg_amd import aggregation
    A = pyamg.gallery.poisson((256, 256, 256), format="csr")
    np.random.seed(1)
    with aggregation.device_setup(pyamg):
        ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, max_levels=4)
    d = {}
    for i, L in enumerate(ml.levels[:3]):
        M = L.A.tocsr()
        d[f"data{i}"], d[f"indices{i}"], d[f"indptr{i}"] = M.data, M.indices, M.indptr
    np.savez(cache, **d)
z = np.load(cache)
for li in levels:
    A = sp.csr_array((z[f"data{li}"], z[f"indices{li}"], z[f"indptr{li}"]))
    n = A.shape[0]
    dA = DeviceMatrix(sparse_op(A))
    dA.tune(gs_order=1, **tune)
    rng = np.random.RandomState(li)
    dx, db = capi.DeviceArray.from_host(rng.rand(n)), capi.DeviceArray.from_host(rng.rand(n))
    for _ in range(2):
        dA.gauss_seidel(dx, db, sweep="forward")
    capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(4):
        dA.gauss_seidel(dx, db, sweep="forward")
    e1.record(); e1.synchronize()
    print("level", li, n, A.nnz, f"{e0.elapsed_ms(e1) / 4:.4f} ms", dA.lane_info(0), flush=True)
    dA.free()
