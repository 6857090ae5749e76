#!/usr/bin/env python3
"""Design aid: timing model of the tiled sweep (tools/tile_sim.cpp) on the level operators of a Poisson SA hierarchy."""
import argparse, ctypes, subprocess, sys, time
from pathlib import Path
import numpy as np
import scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[128, 128, 128])
ap.add_argument("--levels", type=int, nargs="*", default=[0, 1, 2])
ap.add_argument("--tiles", type=int, nargs="*", default=[0])
ap.add_argument("--modes", type=int, nargs="*", default=[0])
ap.add_argument("--hop", type=float, default=2.5)
ap.add_argument("--c0", type=float, default=0.42)
ap.add_argument("--c1", type=float, default=0.011)
ap.add_argument("--W", type=int, default=2048)
ap.add_argument("--cap", type=int, default=512)
ap.add_argument("--backward", type=int, default=0)
ap.add_argument("--rows", type=int, default=64)
ap.add_argument("--c2", type=float, default=0.0)
ap.add_argument("--nolimit", type=int, default=0)
a = ap.parse_args()
cache = Path("/tmp") / ("hier_" + "x".join(map(str, a.grid)) + ".npz")
if not cache.exists():
    import oracle.refimport  # noqa
    import pyamg
    t = time.time()
    A = pyamg.gallery.poisson(a.grid, format="csr")
    np.random.seed(1)
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
    d = {}
    for i, L in enumerate(ml.levels[:-1]):
        M = sp.csr_matrix(L.A)
        d[f"p{i}"], d[f"j{i}"] = M.indptr.astype(np.int32), M.indices.astype(np.int32)
    np.savez(cache, **d)
    print(f"setup {time.time()-t:.1f}s")
Z = np.load(cache)
so = Path("/tmp/tile_sim_nolimit.so" if a.nolimit else "/tmp/tile_sim.so")
src = ROOT / "tools" / "tile_sim.cpp"
hdr = ROOT / "pyamg_amd" / "csrc" / "pamg_tile_plan.h"
if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + (["-DSIM_NOLIMIT"] if a.nolimit else []) + [str(src), "-o", str(so), "-lpthread"], check=True)
lib = ctypes.CDLL(str(so))
for li in a.levels:
    if f"p{li}" not in Z:
        continue
    Ap, Aj = np.ascontiguousarray(Z[f"p{li}"]), np.ascontiguousarray(Z[f"j{li}"])
    n = len(Ap) - 1
    for mode in a.modes:
        for G in a.tiles:
            out = np.zeros(16)
            start, stop, step = (n - 1, -1, -1) if a.backward else (0, n, 1)
            nl_guess = 0
            t = time.time()
            rc = lib.tile_sim(n, Ap.ctypes.data_as(ctypes.c_void_p), Aj.ctypes.data_as(ctypes.c_void_p), start, stop, step, G if G > 0 else -1, a.W, a.cap, a.rows,
                              ctypes.c_double(a.c0), ctypes.c_double(a.c1), ctypes.c_double(a.c2), ctypes.c_double(a.hop), mode, out.ctypes.data_as(ctypes.c_void_p), None)
            print(f"L{li} n={n} nnz={Ap[-1]} mode={mode} G={int(out[5])} rc={rc}: span {out[0]/1000:.3f} ms  steps {int(out[1])} levels {int(out[2])} "
                  f"crit.crossings {int(out[3])} busiest tile {out[4]/1000:.3f} ms ideal(levels*c) {out[8]/1000:.3f} ms  glob {int(out[6])} loc {int(out[7])}  [{time.time()-t:.1f}s]", flush=True)
