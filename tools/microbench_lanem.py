#!/usr/bin/env python3
"""The MERGED lane-parallel Gauss-Seidel sweep (pamg_lanem_plan.h: s dependency levels eliminated into one super-level) on the levels of the
SA hierarchy of 3-D Poisson: forward sweep time by s (tune key 33) x look-ahead (key 34) x gate (key 28) against the unmerged fast order;
every variant is compared with the order-exact device sweep (= the reference's bits) after a symmetric sweep.  Not product code."""
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

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[256, 256, 256])
ap.add_argument("--levels", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--s", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--ahead", type=int, nargs="+", default=[23])
ap.add_argument("--grids", type=int, nargs="+", default=[0])
ap.add_argument("--gate", type=int, nargs="+", default=[1])
ap.add_argument("--xcd", type=int, nargs="+", default=[0])
ap.add_argument("--rpw", type=int, nargs="+", default=[0])
ap.add_argument("--sweep", default="forward")
ap.add_argument("--tag", default="lanem")
a = ap.parse_args()
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
np.random.seed(1)
t = time.time()
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
print(f"setup {time.time() - t:.1f}s", flush=True)
spec = extract(ml)
out = []
outp = ROOT / "gpurun_out" / f"microbench_{a.tag}.json"
outp.parent.mkdir(exist_ok=True)


def timeit(fn, reps=5):
    fn(); capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps


for li in a.levels:
    if li >= len(spec.levels) - 1:
        continue
    op = spec.levels[li].A
    n = op.shape[0]
    rng = np.random.RandomState(li)
    x0, b = rng.rand(n), rng.rand(n)
    dA = DeviceMatrix(op)
    dx, db = capi.DeviceArray.from_host(x0), capi.DeviceArray.from_host(b)
    dA.tune(gs_order=0)
    dA.gauss_seidel(dx, db, sweep="symmetric")
    ref = dx.download()
    t_exact = timeit(lambda: dA.gauss_seidel(dx, db, sweep="forward"))
    rec = {"level": li, "rows": int(n), "nnz": int(op.nnz), "dependency_levels": dA.info()["gs_levels_fwd"], "exact_ms": round(t_exact, 4), "variants": []}
    dA.tune(gs_order=1, lane_wide=1)
    for s, rpw in [(s_, r_) for s_ in a.s for r_ in a.rpw]:
        t0 = time.time()
        dA.tune(lane_merge=s, lanem_rpw=rpw)
        for gate in a.gate:
            for ah in a.ahead:
                for G, xcd in [(G_, x_) for G_ in a.grids for x_ in a.xcd]:
                    dA.tune(lanem_ahead=ah, lane_G=G, lane_flags=gate, gran_xcd=xcd)
                    dx.upload(x0)
                    dA.gauss_seidel(dx, db, sweep="symmetric")
                    got = dx.download()
                    tb = time.time() - t0
                    err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
                    ms = timeit(lambda: dA.gauss_seidel(dx, db, sweep=a.sweep))
                    wh_ = 1 if a.sweep == "backward" else 0
                    mi, li_ = dA.lanem_info(wh_), dA.lane_info(wh_)
                    hops = mi["super_levels"] or rec["dependency_levels"]
                    v = {"s": s, "rpw": rpw, "gate": gate, "ahead10": ah, "lane_G": G, "gran_xcd": xcd, "ms_forward": round(ms, 4), "hand_offs": int(hops), "us_per_hand_off": round(1e3 * ms / hops, 3),
                         "max_rel_diff_vs_exact_symmetric_sweep": err, "units_per_row": round(mi["units"] / max(1, mi["rows"]), 3) if mi["rows"] else None,
                         "operands_per_row": round((mi["early_operands"] + mi["old_operands"] + mi["b_operands"]) / max(1, mi["rows"]), 2) if mi["rows"] else None,
                         "slot_GB": round(mi["units"] * 64 * 12 / 1e9, 3) if mi["rows"] else round(li_["entry_slots"] * 12 / 1e9, 3),
                         "grid": mi["launch_grid"] or li_["launch_grid"], "growth": mi["max_growth"], "closed_len_growth": [mi["closed_by_length"], mi["closed_by_growth"]],
                         "first_build_and_sweep_s": round(tb, 2), "timeout": bool(dA.flow_error())}
                    t0 = time.time()
                    rec["variants"].append(v)
                    print(li, json.dumps(v), flush=True)
    out.append(rec)
    outp.write_text(json.dumps(out, indent=1))
    dA.free(); dx.free(); db.free()
print("done")
