#!/usr/bin/env python3
"""Fast-order (lane-parallel) block Gauss-Seidel against the order-exact block kernels on the levels of the 3-D elasticity SA hierarchy
(bench.py's c5): lanes per block row x persistent grid x gate x one-XCD form.  Every variant is checked against the exact sweep (max relative
difference after a symmetric sweep).  Not product code."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401
import pyamg
from pyamg_amd import _capi as capi
from pyamg_amd.aggregation import device_setup
from pyamg_amd.hierarchy import extract
from pyamg_amd.multilevel import DeviceMatrix
from tools.problems import elasticity3d

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--levels", type=int, nargs="+", default=[0, 1, 2])
ap.add_argument("--tag", default="blane")
ap.add_argument("--exp", default="a")
a = ap.parse_args()
A, B = elasticity3d(a.grid)
np.random.seed(1)
t = time.time()
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, B=B, smooth="jacobi", max_coarse=10)
print(f"setup {time.time() - t:.1f}s", flush=True)
spec = extract(ml)
out = []
outp = ROOT / "gpurun_out" / f"microbench_{a.tag}.json"
outp.parent.mkdir(exist_ok=True)


def timeit(fn, reps=8):
    fn(); capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps


def variants_for(li):
    v = [("exact_default", dict(gs_order=0)), ("fast_auto", dict(gs_order=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1))]
    if a.exp == "a":
        v.append(("fast_nogate", dict(lane_flags=0)))
        for G in (16, 32, 64, 128, 256, 512, 1024):
            v.append((f"fast_chip_G{G}", dict(lane_flags=1, lane_G=G, gran_xcd=2)))
        for G in (8, 16, 32, 64):
            v.append((f"fast_xcd_G{G}", dict(lane_G=G, gran_xcd=1)))
        for L in (16, 32, 64):
            v.append((f"fast_L{L}_chip", dict(lane_L=L, lane_G=0, gran_xcd=2)))
    if a.exp == "abl":
        v = [("fast_auto", dict(gs_order=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1))]
    return v


for li in a.levels:
    if li >= len(spec.levels) - 1:
        continue
    op = spec.levels[li].A
    bs = op.blocksize[0]
    n = op.shape[0]; nb = n // bs
    M = op.to_scipy() if hasattr(op, "to_scipy") else op
    rng = np.random.RandomState(li)
    x, b = rng.rand(n), rng.rand(n)
    indptr, indices, data = np.asarray(M.indptr), np.asarray(M.indices), np.asarray(M.data)
    rows = np.repeat(np.arange(nb), np.diff(indptr))
    dm = rows == indices
    Dinv = np.zeros((nb, bs, bs)); Dinv[rows[dm]] = np.linalg.pinv(data[dm])
    print(f"level {li}: {nb} block rows of {bs}x{bs}, {len(indices)} blocks, longest block row {int(np.diff(indptr).max())}", flush=True)
    dA = DeviceMatrix(op)
    db, dx, dD = capi.DeviceArray.from_host(b), capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(Dinv.reshape(-1))
    ref = None
    for name, kw in variants_for(li):
        try:
            dA.tune(**kw)
            dx.upload(x)
            dA.block_gauss_seidel(dx, db, dD, sweep="symmetric")
            capi.sync()
            got = dx.download()
            if ref is None:
                ref = got
            diff = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
            err = dA.flow_error()
            ms = timeit(lambda: dA.block_gauss_seidel(dx, db, dD, sweep="forward"))
            inf = dA.info()
            li_ = dA.lane_info(0)
            rec = {"level": li, "n": n, "bs": bs, "variant": name, "fwd_ms": round(ms, 4), "max_rel_diff_vs_exact": diff, "timeout": err,
                   "levels": inf["gs_levels_fwd"], "us_per_level": round(1e3 * ms / max(inf["gs_levels_fwd"], 1), 3),
                   "lane": {k: li_[k] for k in ("lanes_per_row", "slots_per_lane", "groups", "widest_level_groups", "launch_grid")} if name != "exact_default" else None}
            if name == "fast_auto":
                dA.tune(gs_prof=1)
                dx.upload(x)
                dA.block_gauss_seidel(dx, db, dD, sweep="forward")
                capi.sync()
                pr = dA.lane_profile(0)
                dA.tune(gs_prof=0)
                if len(pr):
                    tail = (pr[:, 2] - pr[:, 1]) * 10          # ns: last operand seen -> published
                    wait = (pr[:, 1] - pr[:, 0]) * 10          # ns: group started -> last operand seen
                    span = (pr[:, 2].max() - pr[:, 0].min()) * 1e-5
                    rec["prof"] = {"tail_ns_median": float(np.median(tail)), "tail_ns_p10": float(np.percentile(tail, 10)), "tail_ns_p90": float(np.percentile(tail, 90)),
                                   "wait_ns_median": float(np.median(wait)), "span_ms": float(span)}
                    np.savez_compressed(ROOT / "gpurun_out" / f"blane_prof_level{li}.npz", prof=pr, indptr=indptr, indices=indices)
        except Exception as e:  # noqa: BLE001
            rec = {"level": li, "variant": name, "error": repr(e)[:300]}
        print(rec, flush=True)
        out.append(rec)
        outp.write_text(json.dumps(out, indent=1))
    dA.free()
