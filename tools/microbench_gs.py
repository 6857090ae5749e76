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
ap.add_argument("--prof", type=int, default=1)
ap.add_argument("--grids", type=int, nargs="*", default=[128, 384])
ap.add_argument("--xgrids", type=int, nargs="*", default=[])
ap.add_argument("--fine-only", type=int, default=0, help="level 0 only, no hierarchy setup")
a = ap.parse_args()
A = pyamg.gallery.poisson(a.grid, format="csr")
np.random.seed(1)
t = time.time()
if a.fine_only:
    from pyamg_amd.hierarchy import sparse_op
    class _L:  # noqa: E701
        pass
    L0 = _L(); L0.A = sparse_op(A)
    class _S:  # noqa: E701
        pass
    spec = _S(); spec.levels = [L0, None]
else:
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
    print(f"setup {time.time() - t:.1f}s levels={len(ml.levels)}", flush=True)
    spec = extract(ml)
out = []
od_early = ROOT / "gpurun_out"
od_early.mkdir(exist_ok=True)
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
    variants = [("auto", dict(gs_mode=0, gran_xcd=0, gran_cap=0, gs_prof=0)), ("launch", dict(gs_mode=1)), ("single", dict(gs_mode=3))]
    variants += [(f"gran_G{G}", dict(gs_mode=2, gran_cap=G, gran_xcd=2, gs_prof=0)) for G in ([0] + a.grids)]
    variants += [(f"granxcd_G{G}", dict(gs_mode=2, gran_cap=G, gran_xcd=1, gs_prof=0)) for G in [0] + a.xgrids]
    if a.prof:
        variants += [("granprof", dict(gs_mode=2, gran_cap=0, gran_xcd=2, gs_prof=1))]
    for name, kw in variants:
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
        if kw.get("gs_prof"):
            pr = dA.gs_profile(0)
            if len(pr) and a.fine_only:
                np.save(od_early / f"prof_{a.tag}_{name}.npy", pr)
            if len(pr):
                lev = pr[:, 7]
                nl = int(lev.max()) + 1
                fin = np.zeros(nl); polled = np.zeros(nl); staged = np.zeros(nl); arrive = np.zeros(nl)
                for l in range(nl):
                    m = lev == l
                    fin[l] = pr[m, 4].max(); polled[l] = pr[m, 2].max(); staged[l] = pr[m, 3].max(); arrive[l] = pr[m, 0].max()
                t = 0.01   # us per tick
                hop = np.diff(fin) * t
                wait = (polled[1:] - fin[:-1]) * t        # previous level finished -> this level's data seen (wave 0)
                stg = (staged[1:] - polled[1:]) * t       # -> all waves staged
                rowp = (fin[1:] - staged[1:]) * t         # -> row phase + stores issued
                slack = (fin[:-1] - arrive[1:]) * t       # how early the workgroup arrived
                q = lambda v: [round(float(np.percentile(v, p)), 2) for p in (10, 50, 90)]
                rec[name]["prof_us_p10_50_90"] = {"hop": q(hop), "prev_fin_to_polled": q(wait), "polled_to_staged": q(stg),
                                                  "staged_to_fin": q(rowp), "arrived_before_prev_fin": q(slack),
                                                  "span_ms": round(float((fin[-1] - pr[:, 0].min()) * t / 1000), 4)}
                print("   prof", rec[name]["prof_us_p10_50_90"], flush=True)
    out.append(rec)
    dA.free()
od = ROOT / "gpurun_out"
od.mkdir(exist_ok=True)
(od / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
