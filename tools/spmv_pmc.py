#!/usr/bin/env python3
"""Fine-level residual SpMV r = b - A x alone (the kernel bench.py's roofline is quoted on), a few launches:
the command bench.py runs under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to count the kernel's HBM traffic in
the same run as the timing.  Usage: spmv_pmc.py NX [NY [NZ]] [--launches K]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401  (binds libamdhip64 before libpyamg_amd.so, as bench.py does)
from pyamg_amd import _capi as capi  # noqa: E402
from pyamg_amd.hierarchy import sparse_op  # noqa: E402
from pyamg_amd.multilevel import DeviceMatrix  # noqa: E402
from tools.problems import poisson_csr  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
launches = 12
for a in sys.argv[1:]:
    if a.startswith("--launches="):
        launches = int(a.split("=", 1)[1])
grid = tuple(int(a) for a in args)
A = poisson_csr(grid)
n = A.shape[0]
dA = DeviceMatrix(sparse_op(A))
rng = np.random.RandomState(0)
x = capi.DeviceArray.from_host(rng.rand(n))
b = capi.DeviceArray.from_host(rng.rand(n))
r = capi.DeviceArray(n, np.float64)
for _ in range(launches):
    dA.spmv(capi.SPMV_RESID, x, r, b=b)
capi.sync()
print("ok", n, A.nnz, launches)
