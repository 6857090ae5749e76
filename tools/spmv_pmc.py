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
idx16 = None
for a in sys.argv[1:]:
    if a.startswith("--idx16="):
        idx16 = int(a.split("=", 1)[1])
if idx16 is not None:
    dA.tune(idx16=idx16)
opt = {k: int(v) for k, v in (a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a) if k in ("val8", "flags", "cap", "rowpat", "kz")}
if "cap" in opt:
    dA.tune(lds_entries=opt["cap"])
if "flags" in opt:
    dA.tune(stream_flags=opt["flags"])
if "val8" in opt:
    dA.tune(val8=opt["val8"])
if "rowpat" in opt:
    dA.tune(rowpat=opt["rowpat"])
if "kz" in opt:
    dA.tune(rowmask_kz=opt["kz"])
if any(a == "--general=1" for a in sys.argv[1:]):
    # the general kernel on the CSR arrays (bench.py's roofline lead): no value codes, no row forms, LDS window 1536
    dA.tune(val8=0, rowgather=0, lds_entries=1536, max_rows=1024)
for _ in range(launches):
    dA.spmv(capi.SPMV_RESID, x, r, b=b)
capi.sync()
e0, e1 = capi.Event(), capi.Event()
e0.record()
for _ in range(launches):
    dA.spmv(capi.SPMV_RESID, x, r, b=b)
e1.record()
e1.synchronize()
ms = e0.elapsed_ms(e1) / launches
by = 12 * A.nnz + 4 * (n + 1) + 24 * n
print("ok", n, A.nnz, launches, f"idx16={idx16} {opt} codes={dA.value_codes()} resid {ms:.4f} ms  {by / ms / 1e6:.1f} GB/s algorithmic ({100 * by / ms / 1e6 / 8000:.1f} % of 8 TB/s)")
