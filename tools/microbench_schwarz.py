#!/usr/bin/env python3
"""Multiplicative overlapping Schwarz on the device: ONE persistent launch per sweep (round 6) against one launch per dependency level (rounds 3 - 5)
and against the reference's sequential sweep on one host core (oracle/_ref's amg_core when it travelled, else the C restatement of oracle/).

  kernel level : pamg_schwarz_sweep on a 2-D / 3-D Poisson operator, subdomain of row i = the pattern of row i (the reference's default), both modes,
                 bit-compared with each other
  cycle level  : a smoothed-aggregation hierarchy with Schwarz pre / post smoothers built by the reference (when oracle/_ref is there), V-cycles on the
                 device; the scheduler of THIS process is PAMG_SCHWARZ_LEVELS (run the tool twice)
Not product code."""
import argparse, ctypes as C, json, os, sys, time
from pathlib import Path
import numpy as np
import scipy.sparse as sp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyamg_amd import _capi as capi
from pyamg_amd import relaxation as grelax
from pyamg_amd.hierarchy import sparse_op
from pyamg_amd.multilevel import DeviceMatrix, DeviceMultilevelSolver
from tools.problems import poisson_csr

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[512, 512])
ap.add_argument("--cycle-grid", type=int, nargs="+", default=[384, 384])
ap.add_argument("--tag", default="schwarz")
ap.add_argument("--no-kernel", action="store_true")
ap.add_argument("--no-cycle", action="store_true")
a = ap.parse_args()
out = {"scheduler_of_this_process": "one launch per dependency level" if os.environ.get("PAMG_SCHWARZ_LEVELS", "0") not in ("", "0") else "one persistent launch per sweep"}
lib = capi.lib()
i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)       # noqa: E731

if not a.no_kernel:
    A = sp.csr_matrix(poisson_csr(tuple(a.grid)))
    A.sort_indices()
    n = A.shape[0]
    t0 = time.perf_counter()
    sub, sptr, inv, iptr = grelax.schwarz_parameters(A)
    t_par = time.perf_counter() - t0
    Sp, Sj, Tp, Tx = i32(sptr), i32(sub), i32(iptr), np.ascontiguousarray(inv, dtype=np.float64)
    dA = DeviceMatrix(sparse_op(A))
    h = C.c_void_p()
    capi.check(lib.pamg_schwarz_create(C.byref(h), dA.handle, n, capi.ptr(Sp), capi.ptr(Sj), capi.ptr(Tp), capi.ptr(Tx)), "pamg_schwarz_create")
    rng = np.random.RandomState(0)
    x0, b = rng.rand(n), rng.rand(n)
    bd = capi.DeviceArray.from_host(b)
    res, got = {}, {}
    for mode, name in ((0, "persistent"), (1, "level_launches")):
        capi.check(lib.pamg_schwarz_set_mode(h, mode), "set_mode")
        xd = capi.DeviceArray.from_host(x0)
        capi.check(lib.pamg_schwarz_sweep(h, xd.ptr, bd.ptr, 0, n, 1, None), "sweep")        # builds the schedule
        capi.check(lib.pamg_schwarz_sweep(h, xd.ptr, bd.ptr, n - 1, -1, -1, None), "sweep")
        capi.sync()
        xd = capi.DeviceArray.from_host(x0)
        e0, e1 = capi.Event(), capi.Event()
        reps = 10
        e0.record()
        for _ in range(reps):
            capi.check(lib.pamg_schwarz_sweep(h, xd.ptr, bd.ptr, 0, n, 1, None), "sweep")
            capi.check(lib.pamg_schwarz_sweep(h, xd.ptr, bd.ptr, n - 1, -1, -1, None), "sweep")
        e1.record(); e1.synchronize()
        res[name + "_ms_per_sweep"] = round(e0.elapsed_ms(e1) / (2 * reps), 4)
        err = C.c_int(0)
        capi.check(lib.pamg_schwarz_error(h, C.byref(err)), "error")
        res[name + "_error_word"] = int(err.value)
        got[name] = xd.download()
    info = (C.c_int64 * 4)()
    capi.check(lib.pamg_schwarz_info(h, info), "info")
    # the reference's sweep on one host core
    from oracle import oracle as orc
    xc = x0.copy()
    t0 = time.perf_counter()
    for _ in range(2):
        orc.overlapping_schwarz_csr(i32(A.indptr), i32(A.indices), A.data, xc, b, Tx, Tp, Sj, Sp, 0, n, 1)
        orc.overlapping_schwarz_csr(i32(A.indptr), i32(A.indices), A.data, xc, b, Tx, Tp, Sj, Sp, n - 1, -1, -1)
    t_cpu = (time.perf_counter() - t0) / 4
    out["kernel"] = {"grid": a.grid, "n": n, "subdomains": int(info[0]), "largest_subdomain": int(info[1]), "dependency_levels_fwd_bwd": [int(info[2]), int(info[3])],
                     **res, "bit_identical_between_schedulers": bool(np.array_equal(got["persistent"], got["level_launches"])),
                     "speedup_persistent_over_level_launches": round(res["level_launches_ms_per_sweep"] / res["persistent_ms_per_sweep"], 2),
                     "us_per_dependency_level_persistent": round(1e3 * res["persistent_ms_per_sweep"] / max(1, int(info[2])), 3),
                     "us_per_dependency_level_level_launches": round(1e3 * res["level_launches_ms_per_sweep"] / max(1, int(info[2])), 3),
                     "one_host_core_ms_per_sweep": round(1e3 * t_cpu, 2), "host_schwarz_parameters_s": round(t_par, 2)}
    print(json.dumps(out["kernel"]), flush=True)
    lib.pamg_schwarz_destroy(h)
    dA.free()

if not a.no_cycle:
    import oracle.refimport as ri
    if ri.available():
        import pyamg
        A = pyamg.gallery.poisson(tuple(a.cycle_grid), format="csr")
        np.random.seed(3)
        t0 = time.perf_counter()
        ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=("schwarz", {"sweep": "symmetric"}), postsmoother=("schwarz", {"sweep": "symmetric"}))
        t_setup = time.perf_counter() - t0
        dml = DeviceMultilevelSolver(ml)
        n = A.shape[0]
        b, x0 = np.random.rand(n), np.random.rand(n)
        k = 6
        r_ref, r_gpu = [], []
        t0 = time.perf_counter()
        ml.solve(b, x0=x0, tol=1e-30, maxiter=k, residuals=r_ref)
        t_ref = (time.perf_counter() - t0) / k
        dml.solve(b, x0=x0, tol=1e-30, maxiter=k, residuals=r_gpu)
        xd, bd = capi.DeviceArray.from_host(x0), capi.DeviceArray.from_host(b)
        dml.load_device(xd, bd)
        dml.iterate_device(3, want_residuals=False)
        capi.sync()
        t0 = time.perf_counter()
        dml.iterate_device(20, want_residuals=False)
        capi.sync()
        ms = (time.perf_counter() - t0) * 1e3 / 20
        out["cycle"] = {"grid": a.cycle_grid, "n": n, "levels": [int(L.A.shape[0]) for L in dml.levels], "ms_per_cycle": round(ms, 3),
                        "reference_one_core_ms_per_cycle": round(1e3 * t_ref, 1), "speedup_over_reference": round(1e3 * t_ref / ms, 1),
                        "residual_parity_max_rel": float(np.max(np.abs(np.array(r_gpu) - np.array(r_ref)) / np.array(r_ref))), "reference_setup_s": round(t_setup, 1),
                        "stats": dml.stats()}
        print(json.dumps(out["cycle"]), flush=True)
    else:
        out["cycle"] = "oracle/_ref not on this box"
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
