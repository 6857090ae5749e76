#!/usr/bin/env python3
"""Fast-order (lane-parallel) Gauss-Seidel against the order-exact schedulers on the levels of the 256^3 SA hierarchy:
lane width x persistent grid x one-XCD form.  Every variant is checked against the exact sweep (max relative difference
after a symmetric sweep).  Not product code."""
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
ap.add_argument("--levels", type=int, nargs="+", default=[3, 2, 1, 0])
ap.add_argument("--tag", default="lane")
ap.add_argument("--exp", default="a")
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


def variants_for(li, n):
    v = [("exact_default", dict(gs_order=0))]
    if a.exp == "a":
        v.append(("fast_auto", dict(gs_order=1, lane_wide=1, lane_L=0, lane_G=0, gran_xcd=0)))
        if li == 0:
            for L in (8,):
                v.append((f"fast_L{L}", dict(lane_L=L)))
            for G in (512, 1024, 1536):
                v.append((f"fast_L4_G{G}", dict(lane_L=0, lane_G=G)))
        else:
            for L in (16, 32, 64):
                v.append((f"fast_L{L}", dict(lane_L=L, lane_G=0)))
            v.append(("fast_auto_xcd", dict(lane_L=0, lane_G=0, gran_xcd=1)))
            v.append(("fast_auto_noxcd", dict(lane_L=0, lane_G=0, gran_xcd=2)))
            for G in (32, 64, 128, 256, 512, 1024):
                v.append((f"fast_noxcd_G{G}", dict(lane_G=G, gran_xcd=2)))
            for G in (16, 32, 64, 128):
                v.append((f"fast_xcd_G{G}", dict(lane_G=G, gran_xcd=1)))
    if a.exp == "b":                       # gate operand on / off x grid, static and one-XCD forms
        v.append(("fast_auto", dict(gs_order=1, lane_wide=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1)))
        v.append(("fast_auto_nogate", dict(lane_flags=0)))
        if li == 1:
            for fl in (1, 0):
                for G in (64, 96, 128, 192, 256, 384, 512):
                    v.append((f"fast_gate{fl}_G{G}", dict(lane_flags=fl, lane_G=G, gran_xcd=2)))
            for G in (128, 256, 512):
                v.append((f"fast_L64_gate1_G{G}", dict(lane_L=64, lane_flags=1, lane_G=G, gran_xcd=2)))
        elif li in (2, 3):
            for fl in (1, 0):
                for G in (8, 16, 24, 32, 48):
                    v.append((f"fast_xcd_gate{fl}_G{G}", dict(lane_flags=fl, lane_G=G, gran_xcd=1)))
            for G in (16, 32):
                v.append((f"fast_L64_xcd_gate1_G{G}", dict(lane_L=64, lane_flags=1, lane_G=G, gran_xcd=1)))
        elif li == 0:
            for G in (256, 384, 512):
                v.append((f"fast_gate1_G{G}", dict(lane_flags=1, lane_G=G)))
    if a.exp == "c":                       # lanes per row x grid (gate on)
        v.append(("fast_auto", dict(gs_order=1, lane_wide=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1)))
        if li == 1:
            for L, Gs in ((32, (192, 256, 384, 512, 768)), (64, (384, 512, 640, 768, 1024, 1536))):
                for G in Gs:
                    v.append((f"fast_L{L}_G{G}", dict(lane_L=L, lane_G=G, gran_xcd=2)))
        elif li in (2, 3):
            for L, Gs in ((32, (32, 40)), (64, (32, 48, 64, 96))):
                for G in Gs:
                    v.append((f"fast_L{L}_xcd_G{G}", dict(lane_L=L, lane_G=G, gran_xcd=1)))
    if a.exp == "f":                       # deep gate: wide schedules with many waves; static L64 gate on / off
        v.append(("fast_auto", dict(gs_order=1, lane_wide=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1)))
        if li == 0:
            for L in (4, 8):
                for G in (256, 512, 1024, 1536):
                    v.append((f"fast_L{L}_G{G}", dict(lane_L=L, lane_G=G, lane_flags=1)))
            v.append(("fast_L4_G512_nogate", dict(lane_L=4, lane_G=512, lane_flags=0)))
        elif li == 1:
            for fl in (1, 0):
                for G in (384, 512, 768, 1024):
                    v.append((f"fast_gate{fl}_G{G}", dict(lane_flags=fl, lane_G=G)))
    if a.exp == "x":                       # round 5: the one-XCD ticket form on the LARGE level (one XCD reads 1.3 TB/s: profiles/r05_bandwidth_one_xcd.txt)
        v.append(("fast_auto", dict(gs_order=1, lane_wide=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1)))
        for L in (16, 32, 64):
            for G in (64, 128, 256):
                v.append((f"fast_xcd_L{L}_G{G}", dict(lane_L=L, lane_G=G, gran_xcd=1)))
        v.append(("fast_xcd_L16_G256_nogate", dict(lane_L=16, lane_G=256, gran_xcd=1, lane_flags=0)))
    # (exp "y", the one-XCD STATIC form -- a census at the start instead of a ticket per group, tune gran_xcd = 3 -- was measured and removed:
    #  profiles/r05_microbench_lane_one_xcd_static_census_not_kept.json)
    if a.exp == "g":                       # round 5, after the prefetch went: persistent workgroups of the static form on level 1
        v.append(("fast_auto", dict(gs_order=1, lane_wide=1, lane_L=0, lane_G=0, gran_xcd=0, lane_flags=1)))
        for G in (384, 448, 521, 640, 768, 1024):
            v.append((f"fast_G{G}", dict(lane_G=G)))
        v.append(("fast_auto_again", dict(lane_G=0)))
    if a.exp == "h":                       # line-scan form on the grid stencil
        v.append(("tile_exact", dict(gs_order=0)))
        v.append(("lines_auto", dict(gs_order=1, line_scan=1, lane_G=0, lane_flags=1)))
        for G in (128, 256, 512, 1024, 1536):
            v.append((f"lines_G{G}", dict(lane_G=G)))
        for G in (256, 1024):
            v.append((f"lines_nogate_G{G}", dict(lane_G=G, lane_flags=0)))
    return v


for li in a.levels:
    if li >= len(spec.levels) - 1:
        continue
    op = spec.levels[li].A
    n = op.shape[0]
    rng = np.random.RandomState(li)
    x, b = rng.rand(n), rng.rand(n)
    dA = DeviceMatrix(op)
    db, dx = capi.DeviceArray.from_host(b), capi.DeviceArray.from_host(x)
    ref = None
    for name, kw in variants_for(li, n):
        try:
            dA.tune(**kw)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric")
            capi.sync()
            got = dx.download()
            if ref is None:
                ref = got
            diff = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
            err = dA.flow_error()
            ms = timeit(lambda: dA.gauss_seidel(dx, db, sweep="forward"))
            inf = dA.info()
            li_ = dA.lane_info(0)
            ln_ = dA.line_info(0)
            rec = {"level": li, "n": n, "variant": name, "fwd_ms": round(ms, 4), "max_rel_diff_vs_exact": diff, "timeout": err, "line": ln_ if ln_["lines"] else None,
                   "levels": inf["gs_levels_fwd"], "us_per_level": round(1e3 * ms / max(inf["gs_levels_fwd"], 1), 3),
                   "lane": {k: li_[k] for k in ("lanes_per_row", "slots_per_lane", "groups", "widest_level_groups", "launch_grid")} if name != "exact_default" else None}
            if name == "fast_auto":
                dA.tune(gs_prof=1)
                dx.upload(x)
                dA.gauss_seidel(dx, db, sweep="forward")
                capi.sync()
                pr = dA.lane_profile(0)
                dA.tune(gs_prof=0)
                if len(pr):
                    tail = (pr[:, 2] - pr[:, 1]) * 10          # ns: last operand seen -> published
                    wait = (pr[:, 1] - pr[:, 0]) * 10          # ns: group started -> last operand seen
                    span = (pr[:, 2].max() - pr[:, 0].min()) * 1e-5
                    xcds = np.bincount((pr[:, 3] & 15).astype(int), minlength=8).tolist()
                    rec["prof_slab_xcd"] = sorted({(int(blk) % 8, int(xc)) for blk, xc in zip(pr[::997, 3] >> 4, pr[::997, 3] & 15)})
                    rec["prof"] = {"tail_ns_median": float(np.median(tail)), "tail_ns_p90": float(np.percentile(tail, 90)),
                                   "wait_ns_median": float(np.median(wait)), "span_ms": float(span), "groups_by_xcd": xcds}
        except Exception as e:  # noqa: BLE001
            rec = {"level": li, "variant": name, "error": repr(e)[:300]}
        print(rec, flush=True)
        out.append(rec)
        outp.write_text(json.dumps(out, indent=1))
    dA.free()
