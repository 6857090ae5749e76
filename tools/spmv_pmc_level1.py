#!/usr/bin/env python3
"""y = A1 x alone, A1 = the level-1 operator of the 256^3 SA hierarchy (31 entries per row, unsorted SA rows): the command
tools/pmc_stall_probe.py --level1 runs under rocprofv3 --pmc.  Not product code."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import oracle.refimport  # noqa
import pyamg
from pyamg_amd import _capi as capi
from pyamg_amd import aggregation
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix

cache = Path("/tmp/pamg_level1_256.npz")
if cache.exists():
    z = np.load(cache)
    import scipy.sparse as sp
    A1 = sp.csr_array((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
else:
    A = pyamg.gallery.poisson((256, 256, 256), format="csr")
    np.random.seed(1)
    with aggregation.device_setup(pyamg):
        ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, max_levels=2)
    A1 = ml.levels[1].A.tocsr()
    np.savez(cache, data=A1.data, indices=A1.indices, indptr=A1.indptr, shape=np.array(A1.shape))
n = A1.shape[0]
dA = DeviceMatrix(sparse_op(A1))
rng = np.random.RandomState(0)
x = capi.DeviceArray.from_host(rng.rand(n))
y = capi.DeviceArray(n, np.float64)
for _ in range(8):
    dA.spmv(capi.SPMV_SET, x, y)
capi.sync()
e0, e1 = capi.Event(), capi.Event()
e0.record()
for _ in range(8):
    dA.spmv(capi.SPMV_SET, x, y)
e1.record(); e1.synchronize()
print("ok", n, A1.nnz, f"{e0.elapsed_ms(e1) / 8:.4f} ms")
