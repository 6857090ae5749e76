#!/usr/bin/env python3
"""Where the host time of one hierarchy goes (setup through device_setup, then the upload): cProfile of both, top of the cumulative list.  Not product code."""
import cProfile, io, pstats, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401
import pyamg
from pyamg_amd import DeviceMultilevelSolver
from pyamg_amd.aggregation import device_setup

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A = pyamg.gallery.poisson((n, n, n), format="csr")
np.random.seed(1)
with device_setup(pyamg):
    pyamg.smoothed_aggregation_solver(pyamg.gallery.poisson((32, 32, 32), format="csr"), max_coarse=10)     # warm the library
for what in ("setup", "upload"):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    c0 = time.process_time()
    pr.enable()
    if what == "setup":
        with device_setup(pyamg):
            ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=("gauss_seidel", {"sweep": "symmetric"}), postsmoother=("gauss_seidel", {"sweep": "symmetric"}))
    else:
        dml = DeviceMultilevelSolver(ml)
    pr.disable()
    print(f"== {what}: {time.perf_counter() - t0:.2f} s wall, {time.process_time() - c0:.1f} CPU-seconds of this process (all threads)", flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print("\n".join(l[:200] for l in s.getvalue().splitlines()[:75]))
