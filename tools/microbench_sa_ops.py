#!/usr/bin/env python3
"""The SA-level operators of the 3-D Poisson hierarchy (A1, P0, R0: one distinct value per entry, 10-byte stream with 16-bit column codes) in the staged
kernel: streaming flags (bit 0 = nontemporal operator stream, bit 1 = XCD-contiguous range order) x LDS window.  Not product code."""
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
ap.add_argument("--tag", default="sa_ops")
ap.add_argument("--idx16", type=int, default=0)       # 1: 16-bit column codes (where the plan has them) against 32-bit columns (flags + 16 here = tune idx16 0)
ap.add_argument("--pads", type=int, nargs="+", default=None)   # LDS padding sweep (tune key 36) x stream flags 0 / 32 (instantiation)
ap.add_argument("--ablate", type=int, default=0)      # flags: +4 no gather, +8 no row phase (results are wrong by design)
a = ap.parse_args()
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
np.random.seed(1)
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
spec = extract(ml)
ops = [("A1", spec.levels[1].A, capi.SPMV_RESID), ("P0", spec.levels[0].P, capi.SPMV_SET), ("R0", spec.levels[0].R, capi.SPMV_SET)]
out = []
for name, op, epi in ops:
    m, n = op.shape
    dA = DeviceMatrix(op)
    rng = np.random.RandomState(0)
    x = capi.DeviceArray.from_host(rng.rand(n)); b = capi.DeviceArray.from_host(rng.rand(m)); y = capi.DeviceArray(m, np.float64)
    ref = None
    by = 12 * op.nnz + 4 * (m + 1) + 8 * n + 8 * m + (8 * m if epi == capi.SPMV_RESID else 0)
    combos = [(1536, fl, 0) for fl in ((1, 5, 9, 13) if a.ablate else (0, 1, 16 + 0, 16 + 1))] if (a.ablate or a.idx16) else [(c_, f_, 0) for c_ in (1536, 1024, 2048, 3072) for f_ in (0, 1, 2, 3)]
    if a.pads is not None:
        combos = [(1536, fl, pad) for pad in a.pads for fl in (0, 32)]
    for cap, fl, pad in combos:
        if True:
            dA.tune(lds_entries=cap, stream_flags=(fl & 15) | (fl & 32), idx16=0 if (a.idx16 and fl & 16) else 1, lds_pad=pad)
            kw = dict(b=b) if epi == capi.SPMV_RESID else {}
            for _ in range(3):
                dA.spmv(epi, x, y, **kw)
            capi.sync()
            got = y.download()
            if ref is None:
                ref = got
            e0, e1 = capi.Event(), capi.Event()
            e0.record()
            for _ in range(20):
                dA.spmv(epi, x, y, **kw)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_ms(e1) / 20
            rec = {"op": name, "shape": [m, n], "nnz": int(op.nnz), "cap": cap, "flags": fl, "lds_pad": pad, "ms": round(ms, 5), "alg_GBps": round(by / ms / 1e6, 1), "bit_identical": bool(np.array_equal(got, ref))}
            print(rec, flush=True)
            out.append(rec)
    dA.free()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
