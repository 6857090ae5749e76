#!/usr/bin/env python3
"""Where the host-side seconds go: the reference's SA setup under pyamg_amd.aggregation.device_setup (cProfile, top entries by
cumulative time) and the upload of the finished hierarchy (DeviceMultilevelSolver(ml), PAMG_TIMING phases on stderr + cProfile).

    PAMG_TIMING=1 python tools/host_profile.py --grid 256 256 256 --smoother gs [--top 30]
"""
import argparse, cProfile, io, pstats, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401
import pyamg
from pyamg_amd import DeviceMultilevelSolver
from pyamg_amd.aggregation import device_setup

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[128, 128, 128])
ap.add_argument("--smoother", default="gs", choices=["gs", "cheby", "jacobi"])
ap.add_argument("--top", type=int, default=30)
ap.add_argument("--no-upload", action="store_true")
a = ap.parse_args()
sm = {"gs": ("gauss_seidel", {"sweep": "symmetric"}), "cheby": ("chebyshev", {"degree": 3, "iterations": 1}),
      "jacobi": ("jacobi", {"omega": 4.0 / 3.0})}[a.smoother]
A = pyamg.gallery.poisson(tuple(a.grid), format="csr")
# context + kernel load outside the profile
with device_setup(pyamg):
    pyamg.smoothed_aggregation_solver(pyamg.gallery.poisson((16, 16, 16), format="csr"), max_coarse=10)


def report(pr, title):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top)
    print(f"==== {title}\n" + "\n".join(l for l in s.getvalue().splitlines() if l.strip())[:12000], flush=True)


np.random.seed(1)
pr = cProfile.Profile()
t = time.time()
pr.enable()
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm, max_coarse=10)
pr.disable()
print(f"setup {time.time() - t:.2f} s, levels {[(L.A.shape[0], L.A.nnz) for L in ml.levels]}", flush=True)
report(pr, "device_setup")
if not a.no_upload:
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    dml = DeviceMultilevelSolver(ml)
    pr.disable()
    print(f"upload {time.time() - t:.2f} s", flush=True)
    report(pr, "upload")
