#!/usr/bin/env python3
"""What a wavefront step of the level-0 line scan costs: hand-off h and chunk time c from grids with lines of 2 / 4 / 8 / 16 chunks of 64 rows and the
same number of wavefront steps (ny + nz - 1): t_step = h + chunks * c.  7-point Poisson on (nz, ny, nx) grids, forward Gauss-Seidel sweep in fast order.
Not product code."""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyamg_amd import _capi as capi
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix
from tools.problems import poisson_csr
import scipy.sparse as sp

out = []
for (nz, ny, nx) in ((128, 128, 128), (128, 128, 256), (128, 128, 512), (128, 128, 1024), (256, 256, 256), (256, 256, 128), (64, 64, 256), (64, 64, 1024)):
    A = sp.csr_matrix(poisson_csr((nz, ny, nx)))
    n = A.shape[0]
    rng = np.random.RandomState(0)
    x0, bh = rng.rand(n), rng.rand(n)
    b = capi.DeviceArray.from_host(bh)
    got = {}
    # (round 6 A/Bs through this loop, both removed: a variant that requested a chunk's operands one chunk ahead, and the lines in eight strips of the visit
    #  order with one XCD each -- profiles/r06_microbench_line_steps_operands_ahead_not_kept.json, r06_microbench_line_strips_one_xcd_each_not_kept.json)
    for flags, name in ((1, "chip_wide"),):
        dA = DeviceMatrix(sparse_op(A))
        dA.tune(gs_order=1, lane_flags=flags)
        x = capi.DeviceArray.from_host(x0)
        dA.gauss_seidel(x, b, sweep="symmetric")
        got[name] = x.download()
        for _ in range(3):
            dA.gauss_seidel(x, b, sweep="forward")
        capi.sync()
        e0, e1 = capi.Event(), capi.Event()
        reps = 20
        e0.record()
        for _ in range(reps):
            dA.gauss_seidel(x, b, sweep="forward")
        e1.record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / reps
        info = dA.line_info(0)
        steps = ny + nz - 1
        rec = {"grid": [nz, ny, nx], "form": name, "rows": n, "chunks_per_line": (nx + 63) // 64, "wavefront_steps": steps, "ms": round(ms, 4),
               "us_per_step": round(1e3 * ms / steps, 3), "GBps": round((12 * A.nnz + 24 * n) / ms / 1e6, 1), "launch_grid": info.get("launch_grid"),
               "timeout": bool(dA.flow_error())}
        print(rec, flush=True)
        out.append(rec)
        dA.free()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "microbench_line_steps.json").write_text(json.dumps(out, indent=1))
