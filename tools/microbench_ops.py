#!/usr/bin/env python3
"""LDS-window sweep per operator of a real SA hierarchy (A, P, R on every level).  Not product code."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa
import pyamg
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import extract
from pyamg_amd.multilevel import DeviceMatrix
from tools.microbench import timeit

g = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [256, 256, 256]
A = pyamg.gallery.poisson(g, format="csr")
np.random.seed(1)
t = time.time(); ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10); print(f"setup {time.time()-t:.1f}s", flush=True)
spec = extract(ml)
out = {}
rng = np.random.RandomState(0)
for li, L in enumerate(spec.levels[:-1]):
    for nm, op in (("A", L.A), ("P", L.P), ("R", L.R)):
        if op.nnz < 500_000:
            continue
        dM = DeviceMatrix(op)
        dx = capi.DeviceArray.from_host(rng.rand(op.shape[1])); dy = capi.DeviceArray(op.shape[0], np.float64)
        bytes_alg = 12 * op.nnz + 4 * (op.shape[0] + 1) + 8 * op.shape[1] + 8 * op.shape[0]
        rec = {}
        for cap in (512, 1536):
            for fl in (0, 2, 1):
                dM.tune(lds_entries=cap, stream_flags=fl)
                ms = timeit(lambda: dM.spmv(capi.SPMV_SET, dx, dy), 10)
                rec[f"{cap}/f{fl}"] = round(bytes_alg / ms / 1e6, 1)
        best = max(rec, key=rec.get)
        print(f"L{li}.{nm} {op.shape} nnz/row={op.nnz/op.shape[0]:.1f}  GB/s by cap: {rec}  best={best}", flush=True)
        out[f"L{li}.{nm}"] = rec
        dM.free()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_ops.json").write_text(json.dumps(out, indent=1))
