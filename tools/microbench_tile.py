#!/usr/bin/env python3
"""Tiled order-exact Gauss-Seidel sweep: per hierarchy level, the round-1 schedulers against the tiled sweep for
several tilings; every variant is checked bit-for-bit against the oracle.  Not product code (design aid)."""
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
ap.add_argument("--tag", default="tile")
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--tiles", type=int, nargs="*", default=[0])
ap.add_argument("--rings", type=int, nargs="*", default=[0])
ap.add_argument("--slots", type=int, nargs="*", default=[0])
ap.add_argument("--inflight", type=int, nargs="*", default=[-1])
ap.add_argument("--parts", type=int, nargs="*", default=[1])
ap.add_argument("--extra", default="", help="JSON list of tune() keyword dicts measured as further variants")
ap.add_argument("--no-tiles", type=int, default=0)
ap.add_argument("--caps", type=int, nargs="*", default=[0])
ap.add_argument("--levels", type=int, nargs="*", default=None)
ap.add_argument("--baseline", type=int, default=1)
ap.add_argument("--prof", type=int, default=1)
ap.add_argument("--fine-only", type=int, default=0)
ap.add_argument("--save-prof", type=int, default=0)
a = ap.parse_args()
A = pyamg.gallery.poisson(a.grid, format="csr")
np.random.seed(1)
t = time.time()
if a.fine_only:
    from pyamg_amd.hierarchy import sparse_op

    class _L:
        pass
    L0 = _L(); L0.A = sparse_op(A)

    class _S:
        pass
    spec = _S(); spec.levels = [L0, None]
else:
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
    print(f"setup {time.time() - t:.1f}s levels={len(ml.levels)}", flush=True)
    spec = extract(ml)
out = []
od = ROOT / "gpurun_out"
od.mkdir(exist_ok=True)
for li, L in enumerate(spec.levels[:-1]):
    if a.levels is not None and li not in a.levels:
        continue
    op = L.A
    n = op.shape[0]
    rng = np.random.RandomState(li)
    x = rng.rand(n); b = rng.rand(n)
    ref = x.copy()
    if a.check:
        orc.relax_gauss_seidel(op, ref, b, 1, "symmetric")
    t0 = time.time()
    dA = DeviceMatrix(op)
    db = capi.DeviceArray.from_host(b)
    dx = capi.DeviceArray.from_host(x)
    rec = {"level": li, "n": n, "nnz": op.nnz, "fmt": op.fmt}
    variants = []
    if a.baseline:
        variants.append(("round1_auto", dict(gs_mode=0, tile_default=0)))
    for G in a.tiles:
        for W in a.rings:
            for cap in a.caps:
                for D in a.slots:
                    for Q in a.inflight:
                        for pt in a.parts:
                            variants.append((f"tile_G{G}_W{W}_c{cap}_D{D}_Q{Q}_p{pt}", dict(gs_mode=5, tile_G=G, tile_W=W, tile_cap=cap, tile_D=D, tile_Q=Q, tile_part=pt, gs_prof=0)))
    if a.no_tiles:
        variants = [v for v in variants if not v[0].startswith("tile_")]
    for kw in (json.loads(a.extra) if a.extra else []):
        variants.append(("x_" + "_".join(f"{k}{v}" for k, v in kw.items()), dict(tile_default=0, **kw)))
    for name, kw in variants:
        dA.tune(gs_mode=0, gran_xcd=0, gran_cap=0, flow_cap=32)
        dA.tune(**kw)
        dx.upload(x)
        t1 = time.time()
        dA.gauss_seidel(dx, db, sweep="symmetric")
        capi.sync()
        build_s = time.time() - t1
        ok = bool(np.array_equal(dx.download(), ref)) if a.check else None
        err = dA.flow_error()
        info = dA.info()
        ms = timeit(lambda: dA.gauss_seidel(dx, db, sweep="forward"))
        msb = timeit(lambda: dA.gauss_seidel(dx, db, sweep="backward"))
        rec[name] = {"fwd_ms": round(ms, 4), "bwd_ms": round(msb, 4), "exact": ok, "timeout": err, "first_call_s": round(build_s, 2)}
        rec["levels_fwd"] = info["gs_levels_fwd"]
        if kw.get("gs_mode") == 5:
            rec[name]["plan"] = dA.tile_info(0)
            if a.prof:
                dA.tune(gs_prof=1)
                dA.gauss_seidel(dx, db, sweep="forward")
                capi.sync()
                pr = dA.gs_profile(0)
                dA.tune(gs_prof=0)
                if len(pr):
                    tick = 0.01    # us
                    c0, c1, tile, sidx = pr[:, 0].astype(np.float64), pr[:, 1].astype(np.float64), pr[:, 2], pr[:, 3]
                    lo_, gi, gf, gr = (pr[:, k].astype(np.float64) for k in (4, 5, 6, 7))
                    t_start = min(c0[c0 > 0].min(), lo_[lo_ > 0].min())
                    span = (c1.max() - t_start) * tick
                    order = np.lexsort((sidx, tile))
                    c0o, c1o, tileo = c0[order], c1[order], tile[order]
                    same = tileo[1:] == tileo[:-1]
                    wait = ((c0o[1:] - c1o[:-1]) * tick)[same]
                    first = np.r_[True, ~same]
                    last = np.r_[~same, True]
                    tile_start = (c0o[first] - t_start) * tick
                    tile_end = (c1o[last] - t_start) * tick
                    q = lambda v: [round(float(np.percentile(v, p)), 2) for p in (10, 50, 90, 99)]
                    rec[name]["prof"] = {"span_us": round(float(span), 1), "compute_step_us": q((c1 - c0) * tick),
                                         "compute_wait_us": q(wait), "dma_to_gather_us": q((gi - lo_) * tick),
                                         "gather_flight_us": q((gf - gi) * tick), "gather_finish_us": q((gr - gf) * tick),
                                         "ready_to_compute_us": q((c0 - gr) * tick),
                                         "tile_busy_us_p50_max": [round(float(np.median(tile_end - tile_start)), 1), round(float((tile_end - tile_start).max()), 1)],
                                         "tile_first_us_p50_max": [round(float(np.median(tile_start)), 1), round(float(tile_start.max()), 1)],
                                         "steps_per_tile_max": int(np.bincount(tile.astype(np.int64)).max())}
                    if a.save_prof and (li == 0 or a.fine_only):
                        np.save(od / f"prof_{a.tag}_L{li}_{name}.npy", pr)
        print(li, n, name, json.dumps(rec[name]), "levels", info["gs_levels_fwd"], flush=True)
    out.append(rec)
    dA.free()
    (od / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
