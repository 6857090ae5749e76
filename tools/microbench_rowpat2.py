#!/usr/bin/env python3
"""Row-pattern SpMV, one row per lane against two consecutive rows per lane (16-byte accesses): streaming flags x rows per
range, RESID / SET / Jacobi epilogues, every variant against the first one's bits.  Not product code."""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix
from tools.problems import poisson_csr

g = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [256, 256, 256]
tag = sys.argv[4] if len(sys.argv) > 4 else "rowpat2"
A = poisson_csr(tuple(g))
n = A.shape[0]
dA = DeviceMatrix(sparse_op(A))
rng = np.random.RandomState(0)
x, b = capi.DeviceArray.from_host(rng.rand(n)), capi.DeviceArray.from_host(rng.rand(n))
r = capi.DeviceArray(n, np.float64)
w = capi.DeviceArray(n, np.float64)
streamed = 25 * n
out = {"copy_GBps": capi.bandwidth_probe("copy1"), "copy_grid_stride_GBps": capi.bandwidth_probe("copy")}
print(out, flush=True)


def timed(fn, reps=100):
    for _ in range(200):
        fn()
    capi.sync()
    e0, e1 = capi.Event(), capi.Event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / reps


ref = None


def run(name, **tune):
    global ref
    dA.tune(**tune)
    ms = timed(lambda: dA.spmv(capi.SPMV_RESID, x, r, b=b))
    got = r.download()
    if ref is None:
        ref = got
    xj = capi.DeviceArray.from_host(x.download())
    ms_j = timed(lambda: dA.jacobi(xj, b, w, 0.8, iterations=1))
    ms_s = timed(lambda: dA.spmv(capi.SPMV_SET, x, r))
    xj.free()
    out[name] = {"resid_ms": round(ms, 5), "frac_on_streamed_bytes": round(streamed / ms / 1e6 / 8000, 4), "jacobi_ms": round(ms_j, 5), "set_ms": round(ms_s, 5),
                 "same_bits": bool(np.array_equal(ref, got)), "ranges": dA.info()["row_blocks"]}
    print(name, out[name], flush=True)


print("row patterns:", dA.row_patterns(), dA.info(), flush=True)
names = {3: "rowpat", 4: "rowmask", 2: "rowpat2"}
for rp in ([3, 4] if tag.startswith("rowmask") else [3, 2]):
    if rp != 4:
        for fl in (0, 2):
            run(f"{names[rp]}_flags{fl}", stream_flags=fl, rowpat=rp)
    else:
        for fl in (0, 3):
            run(f"rowmask_flags{fl}", rowmask_flags=fl, rowpat=rp)
if tag.startswith("rowmask"):
    for kz in (4, 8):
        for fl in (3, 19):
            run(f"rowmask3d_kz{kz}_flags{fl}", rowmask_flags=fl, rowpat=1, rowmask_kz=kz)
if not tag.startswith("rowmask"):
    for mr, cap in ((256, 2048), (1024, 7168), (2048, 12288)):
        for fl in (0, 2):
            run(f"rowpat2_rows{mr}_flags{fl}", stream_flags=fl, rowpat=2, lds_entries=cap, max_rows=mr)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"microbench_{tag}.json").write_text(json.dumps(out, indent=1))
