#!/usr/bin/env python3
"""Lane mappings of the staged kernel on the operators of a real SA hierarchy (A1, P0, R0 of 256^3): two consecutive entries
per lane (default) against one (lanes on consecutive entries), a few LDS windows.  Not product code."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401
import oracle.refimport  # noqa
import pyamg
from pyamg_amd import _capi as capi
from pyamg_amd import aggregation
from pyamg_amd.hierarchy import extract
from pyamg_amd.multilevel import DeviceMatrix
from tools.microbench import timeit

g = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [256, 256, 256]
A = pyamg.gallery.poisson(g, format="csr")
np.random.seed(1)
t = time.time()
with aggregation.device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
print(f"setup {time.time()-t:.1f}s", flush=True)
spec = extract(ml)
out = {}
rng = np.random.RandomState(0)
for li, L in enumerate(spec.levels[:2]):
    for nm, op in (("A", L.A), ("P", L.P), ("R", L.R)):
        if op is None or op.nnz < 5_000_000 or (li == 0 and nm == "A"):
            continue
        dM = DeviceMatrix(op)
        dx = capi.DeviceArray.from_host(rng.rand(op.shape[1])); dy = capi.DeviceArray(op.shape[0], np.float64)
        rec = {}
        ref = None
        for npl in (2,):
            for cap in (1536, 1024, 768, 2048):
                dM.tune(lds_entries=cap, nnz_per_lane=npl, stream_flags=0)
                ms = timeit(lambda: dM.spmv(capi.SPMV_SET, dx, dy), 10)
                got = dy.download()
                if ref is None:
                    ref = got
                rec[f"npl{npl}/cap{cap}"] = [round(ms, 4), bool(np.array_equal(ref, got))]
        print(f"L{li}.{nm} {op.shape} nnz={op.nnz} nnz/row={op.nnz/op.shape[0]:.1f}  ms: {rec}", flush=True)
        out[f"L{li}.{nm}"] = rec
        dM.free(); dx.free(); dy.free()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_ops2_r03.json").write_text(json.dumps(out, indent=1))
