#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (run through gpurun).  Writes
gpurun_out/microbench_<tag>.json.  Not product code.

  python tools/microbench.py --grid 256 256 256 --sweep         # LDS window / lane-width sweep
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from pyamg_amd import _capi as capi  # noqa: E402
from pyamg_amd.hierarchy import sparse_op  # noqa: E402
from pyamg_amd.multilevel import DeviceMatrix  # noqa: E402
from tools.problems import poisson_csr, spmv_bytes  # noqa: E402


def timeit(fn, reps=20, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs="+", default=[256, 256, 256])
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    out = {"grid": a.grid, "device": None, "results": []}
    lib = capi.lib()
    import ctypes as C
    buf = C.create_string_buffer(256)
    lib.pamg_device_name(0, buf, 256)
    out["device"] = buf.value.decode()
    t = time.time()
    A = poisson_csr(a.grid)
    n = A.shape[0]
    print(f"matrix {a.grid}: n={n} nnz={A.nnz} gen {time.time() - t:.1f}s", flush=True)
    op = sparse_op(A)
    t = time.time()
    dA = DeviceMatrix(op)
    print(f"upload {time.time() - t:.1f}s", flush=True)
    rng = np.random.RandomState(1)
    x = rng.rand(n)
    b = rng.rand(n)
    dx, db = capi.DeviceArray.from_host(x), capi.DeviceArray.from_host(b)
    dy = capi.DeviceArray(n, np.float64)
    dw = capi.DeviceArray(n, np.float64)
    dj = capi.DeviceArray(n, np.float64)
    dout = capi.DeviceArray(1, np.float64)
    bytes_spmv = spmv_bytes(A)
    # copy ceiling: d2d of 2 x 8n... use a 1 GiB copy
    big = capi.DeviceArray(1 << 27, np.float64)
    big2 = capi.DeviceArray(1 << 27, np.float64)
    ms = timeit(lambda: capi.check(lib.pamg_memcpy_d2d(big2.ptr, big.ptr, (1 << 27) * 8, None)), reps=10)
    out["copy_GBps"] = 2 * (1 << 30) / ms / 1e6
    print(f"d2d copy 1 GiB: {ms:.3f} ms -> {out['copy_GBps']:.0f} GB/s (read+write)", flush=True)
    big.free(); big2.free()
    configs = [(2048, 2, 0)]
    if a.sweep:
        configs = [(c, 2, f) for c in (1536, 2048, 3072) for f in (0, 1, 2, 3)]
    ref = A @ x
    for cap, npl, fl in configs:
        dA.tune(lds_entries=cap, nnz_per_lane=npl, stream_flags=fl)
        dA.spmv(capi.SPMV_SET, dx, dy)
        ok = bool(np.array_equal(dy.download(), ref))
        r = {"cap": cap, "npl": npl, "flags": fl, "bit_exact": ok, "row_blocks": dA.info()["row_blocks"]}
        ms = timeit(lambda: dA.spmv(capi.SPMV_SET, dx, dy), a.reps)
        r["spmv_ms"] = ms; r["spmv_GBps"] = bytes_spmv / ms / 1e6
        ms = timeit(lambda: dA.spmv(capi.SPMV_RESID, dx, dy, b=db), a.reps)
        r["resid_ms"] = ms; r["resid_GBps"] = (bytes_spmv + 8 * n) / ms / 1e6
        ms = timeit(lambda: dA.resid_sumsq(dx, db, dout), a.reps)
        r["sumsq_ms"] = ms; r["sumsq_GBps"] = (bytes_spmv) / ms / 1e6
        dj.upload(x)
        ms = timeit(lambda: dA.jacobi(dj, db, dw, 0.7, 2), a.reps)      # 2 sweeps = ping-pong, no copy
        r["jacobi_ms"] = ms / 2; r["jacobi_GBps"] = (bytes_spmv + 8 * n) / (ms / 2) / 1e6
        print(json.dumps(r), flush=True)
        out["results"].append(r)
    # Gauss-Seidel sweep (level scheduled)
    dA.tune(lds_entries=2048, nnz_per_lane=2)
    t = time.time()
    dA.gauss_seidel(dj, db, sweep="symmetric")
    capi.sync()
    out["gs_analysis_s"] = time.time() - t
    info = dA.info()
    ms = timeit(lambda: dA.gauss_seidel(dj, db, sweep="forward"), reps=5, warm=1)
    out["gs_forward_ms"] = ms
    out["gs_levels"] = info["gs_levels_fwd"]
    out["gs_GBps"] = (bytes_spmv + 8 * n + 4 * n) / ms / 1e6
    print(f"GS fwd: {ms:.3f} ms, levels={info['gs_levels_fwd']}, analysis {out['gs_analysis_s']:.1f}s", flush=True)
    od = ROOT / "gpurun_out"
    od.mkdir(exist_ok=True)
    (od / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
