#!/usr/bin/env python3
"""Fine-level residual SpMV of 3-D Poisson: streaming flags (bit 0 non-temporal operator stream, bit 1 XCD-contiguous
row-range order) x LDS window, with the 16-bit column stream.  Not product code."""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix
from tools.problems import poisson_csr

g = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [256, 256, 256]
A = poisson_csr(tuple(g))
n = A.shape[0]
dA = DeviceMatrix(sparse_op(A))
rng = np.random.RandomState(0)
x, b = capi.DeviceArray.from_host(rng.rand(n)), capi.DeviceArray.from_host(rng.rand(n))
r = capi.DeviceArray(n, np.float64)
by = 12 * A.nnz + 4 * (n + 1) + 24 * n
out = {}
for cap in (1536, 1024, 2048, 3072):
    for fl in (0, 1, 2, 3):
        dA.tune(lds_entries=cap, stream_flags=fl)
        for _ in range(5):
            dA.spmv(capi.SPMV_RESID, x, r, b=b)
        capi.sync()
        e0, e1 = capi.Event(), capi.Event()
        e0.record()
        for _ in range(30):
            dA.spmv(capi.SPMV_RESID, x, r, b=b)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 30
        out[f"cap{cap}_flags{fl}"] = {"ms": round(ms, 5), "GBps": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / 8000, 4)}
        print(f"cap{cap} flags{fl}", out[f"cap{cap}_flags{fl}"], flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_spmv_flags_r03.json").write_text(json.dumps(out, indent=1))
