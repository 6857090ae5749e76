#!/usr/bin/env python3
"""Where a row's time goes in the merged lane sweep (tune gs_prof=1: per row arrival, slots arrived, all operands present, published; 10 ns ticks):
medians of the phases, the period of a super-level, how far ahead of the front waves arrive.  Level 1 (default) of the 256^3 SA hierarchy.  Not product code."""
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
ap.add_argument("--level", type=int, default=1)
ap.add_argument("--s", type=int, nargs="+", default=[2, 3])
ap.add_argument("--grids", type=int, nargs="+", default=[768])
ap.add_argument("--rpw", type=int, default=0)
a = ap.parse_args()
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
np.random.seed(1)
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
spec = extract(ml)
op = spec.levels[a.level].A
n = op.shape[0]
rng = np.random.RandomState(0)
dA = DeviceMatrix(op)
dx, db = capi.DeviceArray.from_host(rng.rand(n)), capi.DeviceArray.from_host(rng.rand(n))
dA.tune(gs_order=1, lane_wide=1)
for s in a.s:
    dA.tune(lane_merge=s, lanem_rpw=a.rpw)
    for G in a.grids:
        dA.tune(lane_G=G, gs_prof=0)
        for _ in range(3):
            dA.gauss_seidel(dx, db, sweep="forward")
        capi.sync()
        e0, e1 = capi.Event(), capi.Event()
        e0.record()
        for _ in range(5):
            dA.gauss_seidel(dx, db, sweep="forward")
        e1.record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 5
        dA.tune(gs_prof=1)
        dA.gauss_seidel(dx, db, sweep="forward")
        dA.gauss_seidel(dx, db, sweep="forward")
        capi.sync()
        pr = dA.lane_profile(0)
        lv = dA.lanem_levels(0)
        spins = (pr[:, 0] >> 52) & 4095
        t0 = pr[:, 0] & ((1 << 52) - 1)
        t1, t2, t3 = pr[:, 1], pr[:, 2], pr[:, 3]
        base = t0.min()
        us = lambda v: 0.01 * v
        sup = np.searchsorted(lv, np.arange(pr.shape[0]), side="right") - 1            # stamps are per GROUP (one or two rows)
        # per super-level: when its last row was published
        last_pub = np.zeros(len(lv) - 1)
        np.maximum.at(last_pub, sup, us(t3 - base))
        first_ready = np.full(len(lv) - 1, 1e30)
        np.minimum.at(first_ready, sup, us(t2 - base))
        period = np.diff(last_pub)
        # for every row: how long before the previous super-level was complete did the wave arrive (positive = early)
        prev_done = np.concatenate([[0.0], last_pub[:-1]])[sup]
        lead = prev_done - us(t1 - base)
        out = {"s": s, "grid": G, "ms_forward_unprofiled": round(ms, 4), "total_profiled_ms": round(us(t3.max() - base) / 1e3, 4), "super_levels": int(len(lv) - 1),
               "slots_us_median_p90": [round(float(np.median(us(t1 - t0))), 2), round(float(np.percentile(us(t1 - t0), 90)), 2)],
               "operands_us_median_p90": [round(float(np.median(us(t2 - t1))), 2), round(float(np.percentile(us(t2 - t1), 90)), 2)],
               "tail_us_median_p90": [round(float(np.median(us(t3 - t2))), 2), round(float(np.percentile(us(t3 - t2), 90)), 2)],
               "row_us_median": round(float(np.median(us(t3 - t0))), 2),
               "poll_rounds_median_p90_mean": [float(np.median(spins)), float(np.percentile(spins, 90)), round(float(spins.mean()), 2)],
               "rows_without_repoll_pct": round(100.0 * float((spins == 0).mean()), 1),
               "super_level_period_us_median_p90": [round(float(np.median(period)), 2), round(float(np.percentile(period, 90)), 2)],
               "slots_arrived_before_previous_super_level_done_us_median_p10_p90": [round(float(np.median(lead)), 2), round(float(np.percentile(lead, 10)), 2), round(float(np.percentile(lead, 90)), 2)],
               "ready_minus_previous_done_us_median_p90": [round(float(np.median(us(t2 - base) - prev_done)), 2), round(float(np.percentile(us(t2 - base) - prev_done, 90)), 2)]}
        print(json.dumps(out), flush=True)
