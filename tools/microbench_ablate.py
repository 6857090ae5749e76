#!/usr/bin/env python3
"""Where does the LDS-streamed SpMV spend its time?  Ablations on 3-D Poisson (not product code)."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix
from tools.problems import poisson_csr, spmv_bytes
from tools.microbench import timeit

g = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [256, 256, 256]
A = poisson_csr(g)
n = A.shape[0]
dA = DeviceMatrix(sparse_op(A))
rng = np.random.RandomState(1)
dx = capi.DeviceArray.from_host(rng.rand(n)); db = capi.DeviceArray.from_host(rng.rand(n)); dy = capi.DeviceArray(n, np.float64)
lib = capi.lib()
B = spmv_bytes(A)
out = {}
ms = timeit(lambda: capi.check(lib.pamg_vec_scale(0, n, 1.0000001, dx.ptr, dy.ptr, None)), 20)
out["vec_scale_8B_per_lane_GBps"] = 16 * n / ms / 1e6
ms = timeit(lambda: capi.check(lib.pamg_vec_axpy(0, n, 0.5, dx.ptr, dy.ptr, None)), 20)
out["vec_axpy_GBps"] = 24 * n / ms / 1e6
import os
CAPS = [int(c) for c in os.environ.get('CAPS', '1024,1536,2048').split(',')]
NPLS = [int(c) for c in os.environ.get('NPLS', '2,4').split(',')]
ref = None
for cap in CAPS:
  for npl in NPLS:
    for fl, name in ((0, "full"), (8, "no_row_phase"), (4, "no_gather"), (12, "stream_only")):
        dA.tune(lds_entries=cap, nnz_per_lane=npl, stream_flags=fl)
        name = f"npl{npl}_{name}"
        if fl == 0:
            dA.spmv(capi.SPMV_SET, dx, dy)
            got = dy.download()
            if ref is None:
                ref = A @ dx.download()
            assert np.array_equal(got, ref), (cap, npl)
        ms = timeit(lambda: dA.spmv(capi.SPMV_SET, dx, dy), 20)
        out[f"cap{cap}_{name}"] = {"ms": round(ms, 4), "GBps_alg": round(B / ms / 1e6, 1),
                                  "matrix_stream_GBps": round((12 * A.nnz) / ms / 1e6, 1)}
        print(cap, name, out[f"cap{cap}_{name}"], flush=True)
print(json.dumps({k: v for k, v in out.items() if "vec" in k}))
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_ablate.json").write_text(json.dumps(out, indent=1))


# LDS-staged x windows (tune key 9); ranges capped at 256 rows so that they map one row per lane
dj = capi.DeviceArray.from_host(rng.rand(n)); dw = capi.DeviceArray(n, np.float64)
for cap in (1024, 1536, 2048):
    for xw in (0, 1):
        dA.tune(lds_entries=cap, nnz_per_lane=2, stream_flags=0, max_rows=256)
        dA.spmv(capi.SPMV_SET, dx, dy)
        assert np.array_equal(dy.download(), ref), (cap, xw)
        ms = timeit(lambda: dA.spmv(capi.SPMV_RESID, dx, dy, b=db), 20)
        msj = timeit(lambda: dA.jacobi(dj, db, dw, 0.7, 2), 10) / 2
        print("xwin cap", cap, "on" if xw else "off", "resid ms", round(ms, 4), "GB/s", round((B + 8 * n) / ms / 1e6, 1),
              "| jacobi ms", round(msj, 4), flush=True)
