#!/usr/bin/env python3
"""Order-exact Gauss-Seidel sweep micro-benchmark per hierarchy level: one launch per
dependency level vs the persistent kernel with G workgroups.  Checks every variant
bit-for-bit against the oracle.  Not product code."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: E402,F401
import pyamg  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from pyamg_amd import _capi as capi  # noqa: E402
from pyamg_amd.hierarchy import extract  # noqa: E402
from pyamg_amd.multilevel import DeviceMatrix  # noqa: E402


def timeit(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_ms(e1) / reps


ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[128, 128, 128])
ap.add_argument("--tag", default="gs")
ap.add_argument("--check", type=int, default=1)
a = ap.parse_args()
A = pyamg.gallery.poisson(a.grid, format="csr")
np.random.seed(1)
t = time.time()
ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
print(f"setup {time.time() - t:.1f}s levels={len(ml.levels)}", flush=True)
spec = extract(ml)
out = []
for li, L in enumerate(spec.levels[:-1]):
    op = L.A
    n = op.shape[0]
    rng = np.random.RandomState(li)
    x = rng.rand(n); b = rng.rand(n)
    ref = x.copy()
    if a.check:
        orc.relax_gauss_seidel(op, ref, b, 1, "symmetric")
    dA = DeviceMatrix(op)
    db = capi.DeviceArray.from_host(b)
    dx = capi.DeviceArray.from_host(x)
    rec = {"level": li, "n": n, "nnz": op.nnz, "fmt": op.fmt}
    for name, kw in [("default", dict(gs_mode=0, gran_xcd=0, flow_cap=32, flow_force=0))] + [(f"flowG{G}", dict(gs_mode=0, flow_cap=G, flow_force=1)) for G in (16, 64, 256)] + \
                    [("launch", dict(gs_mode=0, gran_xcd=0, flow_cap=0, flow_force=0)), ("flow1", dict(gs_mode=0, flow_cap=1, flow_force=1))]:
        dA.tune(**kw)
        dx.upload(x)
        dA.gauss_seidel(dx, db, sweep="symmetric")
        capi.sync()
        ok = bool(np.array_equal(dx.download(), ref)) if a.check else None
        err = dA.flow_error()
        info = dA.info()
        ms = timeit(lambda: dA.gauss_seidel(dx, db, sweep="forward"))
        rec[name] = {"fwd_ms": round(ms, 4), "exact": ok, "timeout": err}
        rec["levels_fwd"] = info["gs_levels_fwd"]
        print(li, n, name, rec[name], "levels", info["gs_levels_fwd"], flush=True)
    out.append(rec)
    dA.free()
od = ROOT / "gpurun_out"
od.mkdir(exist_ok=True)
(od / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
