#!/usr/bin/env python3
"""A/B of the general CSR kernel (csr_stream_kernel<double, RESID, npl2>) between library builds inside ONE session (boxes of the pool differ
by up to 7 %): every build runs in its own process (PAMG_LIB), rounds interleaved, best and median of each.

    python tools/ab_csr_stream.py libA.so libB.so [...]   (paths relative to the repo root; 'head' = pyamg_amd/libpyamg_amd.so)
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import sys, json
import numpy as np
sys.path.insert(0, %r)
import torch
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix
from tools.problems import poisson_csr
out = {}
for grid in ((256, 256, 256), (2000, 2000)):
    A = poisson_csr(grid)
    n = A.shape[0]
    rng = np.random.RandomState(0)
    x = capi.DeviceArray.from_host(rng.rand(n)); b = capi.DeviceArray.from_host(rng.rand(n)); r = capi.DeviceArray(n, np.float64)
    by = 12 * A.nnz + 4 * (n + 1) + 24 * n
    for flags in (0, 1):
        dA = DeviceMatrix(sparse_op(A))
        dA.tune(val8=0, rowgather=0, lds_entries=1536, max_rows=1024)
        dA.tune(stream_flags=flags)
        for _ in range(5):
            dA.spmv(capi.SPMV_RESID, x, r, b=b)
        ts = []
        for rep in range(5):
            e0, e1 = capi.Event(), capi.Event()
            e0.record()
            for _ in range(30):
                dA.spmv(capi.SPMV_RESID, x, r, b=b)
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_ms(e1) / 30)
        out["x".join(map(str, grid)) + f" flags={flags}"] = {"best_ms": round(min(ts), 5), "median_ms": round(float(np.median(ts)), 5), "frac_8d_best": round(by / min(ts) / 1e6 / 8000, 4)}
        dA.free()
    for d in (x, b, r):
        d.free()
print("RESULT " + json.dumps(out))
''' % str(ROOT)


def main():
    libs = sys.argv[1:] or ["pyamg_amd/build/ab/libpyamg_amd_r05head.so", "head"]
    res = {}
    for rnd in range(2):
        for lib in libs:
            path = ROOT / ("pyamg_amd/libpyamg_amd.so" if lib == "head" else lib)
            p = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PAMG_LIB=str(path)), capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(lib, "FAILED", p.stderr[-1500:])
                continue
            res.setdefault(lib, []).append(json.loads(line[0][7:]))
            print(f"round {rnd} {lib}: " + json.dumps(res[lib][-1]), flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
