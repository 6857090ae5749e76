#!/usr/bin/env python3
"""Fine-level residual SpMV of 3-D Poisson with 8-bit value codes (tune key 21) against the values streamed as stored:
LDS window x streaming flags, SET / RESID / Jacobi epilogues.  Not product code."""
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
print("value codes:", dA.value_codes(), flush=True)
rng = np.random.RandomState(0)
x, b = capi.DeviceArray.from_host(rng.rand(n)), capi.DeviceArray.from_host(rng.rand(n))
r = capi.DeviceArray(n, np.float64)
w = capi.DeviceArray(n, np.float64)
by = 12 * A.nnz + 4 * (n + 1) + 24 * n
out = {}


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps


ref = None
for v8 in (0, 1):
    for cap in ((1536, 2048) if v8 == 0 else (1536, 1792, 2048, 2304, 3072, 4096, 1024)):
        for fl in ((0, 3) if v8 == 0 else (0, 3)):
            dA.tune(lds_entries=cap, stream_flags=fl, val8=v8)
            ms = timed(lambda: dA.spmv(capi.SPMV_RESID, x, r, b=b))
            got = r.download()
            if ref is None:
                ref = got
            same = bool(np.array_equal(ref, got))
            key = f"val8_{v8}_cap{cap}_flags{fl}"
            out[key] = {"resid_ms": round(ms, 5), "GBps_csr_formula": round(by / ms / 1e6, 1), "frac_csr_formula": round(by / ms / 1e6 / 8000, 4), "same_bits": same}
            print(key, out[key], flush=True)
for v8 in (0, 1):
    dA.tune(lds_entries=1536, stream_flags=0, val8=v8)
    xj = capi.DeviceArray.from_host(rng.rand(n))
    ms_j = timed(lambda: dA.jacobi(xj, b, w, 0.8, iterations=1))
    ms_s = timed(lambda: dA.spmv(capi.SPMV_SET, x, r))
    out[f"val8_{v8}_jacobi_set"] = {"jacobi_ms": round(ms_j, 5), "set_ms": round(ms_s, 5)}
    print(f"val8_{v8}", out[f"val8_{v8}_jacobi_set"], flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_spmv_val8_r03.json").write_text(json.dumps(out, indent=1))
