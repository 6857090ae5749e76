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


def run(tag, **tune):
    global ref
    dA.tune(**tune)
    ms = timed(lambda: dA.spmv(capi.SPMV_RESID, x, r, b=b))
    got = r.download()
    if ref is None:
        ref = got
    xj = capi.DeviceArray.from_host(x.download())
    ms_j = timed(lambda: dA.jacobi(xj, b, w, 0.8, iterations=1))
    ms_s = timed(lambda: dA.spmv(capi.SPMV_SET, x, r))
    xj.free()
    out[tag] = {"resid_ms": round(ms, 5), "frac_csr_formula": round(by / ms / 1e6 / 8000, 4), "jacobi_ms": round(ms_j, 5), "set_ms": round(ms_s, 5),
                "same_bits": bool(np.array_equal(ref, got)), "ranges": dA.info()["row_blocks"]}
    print(tag, out[tag], flush=True)


plan0 = dA.info()
print("default plan of the operator:", plan0, flush=True)
print("row patterns:", dA.row_patterns(), flush=True)
for fl in (0, 1, 2, 3):
    run(f"rowpat_default_plan_flags{fl}", stream_flags=fl, rowpat=1)
for mr, cap in ((256, 2048), (512, 3584), (1024, 7168)):
    run(f"rowpat_rows{mr}", stream_flags=0, rowpat=1, lds_entries=cap, max_rows=mr)
dA.tune(lds_entries=3584, max_rows=512)
for fl in (0, 3):
    run(f"rowgather_default_plan_flags{fl}", stream_flags=fl, rowpat=0)
run("values_as_stored_staged_cap1536", val8=0, rowgather=0, rowpat=0, lds_entries=1536, max_rows=1024, stream_flags=0)
run("codes_staged_cap2048", val8=1, rowgather=0, lds_entries=2048, max_rows=1024)
for cap, mr in ((3584, 512),):
    run(f"rowgather_cap{cap}_rows{mr}", val8=1, rowgather=1, lds_entries=cap, max_rows=mr)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_spmv_val8_r03.json").write_text(json.dumps(out, indent=1))
