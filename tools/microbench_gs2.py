#!/usr/bin/env python3
"""Order-exact Gauss-Seidel on the SA coarse levels of the 256^3 hierarchy: row-range geometry (LDS window = entries per
range, rows per range) x grid size x one-XCD form of the granular sweep.  Every variant must reproduce the default's
sweep bit for bit.  Not product code."""
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
ap.add_argument("--levels", type=int, nargs="+", default=[1, 2])
ap.add_argument("--tag", default="gs2")
ap.add_argument("--exp", default="a")
a = ap.parse_args()
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
np.random.seed(1)
t = time.time()
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
print(f"setup {time.time() - t:.1f}s", flush=True)
spec = extract(ml)


def timeit(fn, reps=5):
    fn(); capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps


out = []
for li in a.levels:
    op = spec.levels[li].A
    n = op.shape[0]
    rng = np.random.RandomState(li)
    x, b = rng.rand(n), rng.rand(n)
    dA = DeviceMatrix(op)
    db, dx = capi.DeviceArray.from_host(b), capi.DeviceArray.from_host(x)
    ref = None
    variants = [("default", {})]
    if a.exp == "c":
        variants = [("transposed", dict(gs_cap=0, _env="1")), ("plain", dict(gs_cap=0, _env="0")), ("transposed_again", dict(gs_cap=0, _env="1")),
                    ("transposed_gcap384", dict(gs_cap=384, _env="1")), ("transposed_gcap768", dict(gs_cap=768, _env="1")),
                    ("transposed_gcap1536", dict(gs_cap=1536, _env="1"))]
    elif a.exp == "b":
        for cap in (256, 384, 512):
            variants.append((f"gcap{cap}", dict(lds_entries=1536, gs_cap=cap, gs_mode=2, gran_xcd=2, gran_cap=0)))
        for cap, G in ((512, 128), (512, 256), (256, 256), (1024, 192)):
            variants.append((f"gcap{cap}_xcd_G{G}", dict(lds_entries=1536, gs_cap=cap, gs_mode=2, gran_xcd=1, gran_cap=G)))
        for cap, G in ((512, 384), (256, 512)):
            variants.append((f"gcap{cap}_G{G}", dict(lds_entries=1536, gs_cap=cap, gs_mode=2, gran_xcd=2, gran_cap=G)))
    else:
        for cap in (512, 768, 1024, 2048):
            variants.append((f"cap{cap}", dict(lds_entries=cap, gs_mode=2, gran_xcd=2)))
        for mr in (16, 32, 64):
            variants.append((f"rows{mr}", dict(lds_entries=1536, max_rows=mr, gs_mode=2, gran_xcd=2)))
        for cap, G in ((768, 512), (512, 768), (1536, 512), (1024, 384)):
            variants.append((f"cap{cap}_G{G}", dict(lds_entries=cap, max_rows=1024, gs_mode=2, gran_xcd=2, gran_cap=G)))
        for cap in (768, 1536):
            variants.append((f"cap{cap}_xcd", dict(lds_entries=cap, max_rows=1024, gs_mode=2, gran_xcd=1, gran_cap=0)))
        variants.append(("tiled", dict(lds_entries=1536, max_rows=1024, gs_mode=5, gran_xcd=0, gran_cap=0)))
    for name, kw in variants:
        try:
            import os
            kw = dict(kw)
            if "_env" in kw:
                os.environ["PAMG_GS_TRANSPOSED"] = kw.pop("_env")
            if kw:
                dA.tune(**kw)
            dx.upload(x)
            dA.gauss_seidel(dx, db, sweep="symmetric")
            capi.sync()
            got = dx.download()
            if ref is None:
                ref = got
            ok = bool(np.array_equal(got, ref))
            err = dA.flow_error()
            ms = timeit(lambda: dA.gauss_seidel(dx, db, sweep="forward"))
            inf = dA.info()
            rec = {"level": li, "n": n, "variant": name, "fwd_ms": round(ms, 4), "exact": ok, "timeout": err, "ranges": inf["row_blocks"],
                   "levels": inf["gs_levels_fwd"], "us_per_level": round(1e3 * ms / max(inf["gs_levels_fwd"], 1), 3)}
        except Exception as e:  # noqa: BLE001
            rec = {"level": li, "variant": name, "error": repr(e)[:200]}
        print(rec, flush=True)
        out.append(rec)
    dA.free()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
