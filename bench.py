#!/usr/bin/env python3
"""Headline benchmark: V-cycle iterations/sec + fine-level SpMV GB/s (% of the MI355X HBM
roofline) -- BASELINE.json's metric -- on one of BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c3j] [--cpu-cycles k]

A *step* is one pass of the hot path: one multigrid V-cycle on the device-resident
hierarchy followed by the convergence-check residual norm ||b - A x|| (exactly one
iteration of the loop in pyamg/multilevel.py:558-569).  Inputs are resident in HBM when the
timed region starts.  The hierarchy is built on the host by the reference itself
(oracle/_ref = the reference compiled from /root/reference; setup stays on the host, north
star) and shipped to HBM once.

Prints ONE JSON line (rank 0):  metric/value/unit/... + "roofline" (fine-level CSR SpMV
kernel, algorithmic bytes / HIP-event time, vs 8 TB/s) + "cpu_baseline" (the reference's
own serial solve on this host, same hierarchy, same b) + "parity" (residual norms of this
run's GPU cycles vs the reference's cycles).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(grid=(2000, 2000), label="gallery.poisson((2000,2000)) SA V-cycle, weighted-Jacobi pre/post, fp64",
               smoother=("jacobi", {"omega": 4.0 / 3.0})),
    # BASELINE.json configs[2] (the north-star's 256^3 problem)
    "c3": dict(grid=(256, 256, 256), label="3D 7-pt Poisson 256^3 SA V-cycle, symmetric Gauss-Seidel, fp64",
               smoother=("gauss_seidel", {"sweep": "symmetric"})),
    # same grid as c3 with the Jacobi smoother (bandwidth-bound variant; not a BASELINE config)
    "c3j": dict(grid=(256, 256, 256), label="3D 7-pt Poisson 256^3 SA V-cycle, weighted-Jacobi pre/post, fp64",
                smoother=("jacobi", {"omega": 4.0 / 3.0})),
    "c4s": dict(grid=(256, 256, 256), label="3D 7-pt Poisson 256^3 SA V-cycle, Chebyshev(3) pre/post, fp64",
                smoother=("chebyshev", {"degree": 3, "iterations": 1})),
    "small": dict(grid=(64, 64, 64), label="3D 7-pt Poisson 64^3 SA V-cycle, symmetric Gauss-Seidel, fp64 (smoke)",
                  smoother=("gauss_seidel", {"sweep": "symmetric"})),
}
SEED = 20260924


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("PAMG_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-cycles", type=int, default=-1, help="reference cycles timed on the host (-1: auto, 0: skip)")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # torch is plumbing only (process group + the contract's cuda synchronize).  It must be
    # imported BEFORE libpyamg_amd.so is loaded: both link libamdhip64.so.7 and torch's copy
    # has to be the one the process binds (measured: the other order loses torch's GPUs).
    import torch
    import numpy as np
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from pyamg_amd import DeviceMultilevelSolver, _capi as capi
    from tools.problems import spmv_bytes
    capi.check(capi.lib().pamg_set_device(local_rank), "set_device")

    import oracle.refimport as ri
    if not ri.available():
        raise SystemExit("bench.py needs the reference build oracle/_ref for the host-side setup phase "
                         "(python oracle/build_ref.py in the build container)")
    import pyamg

    wl = WORKLOADS[args.workload]
    t0 = time.time()
    A = pyamg.gallery.poisson(wl["grid"], format="csr")
    n = A.shape[0]
    np.random.seed(SEED)                       # Arnoldi start vectors of the smoother setup
    ml = pyamg.smoothed_aggregation_solver(A, presmoother=wl["smoother"], postsmoother=wl["smoother"],
                                           max_coarse=10)
    t_setup = time.time() - t0
    np.random.seed(SEED)
    b = np.random.rand(n)
    x0 = np.zeros(n)
    if rank == 0:
        log(f"workload {args.workload}: n={n} nnz={A.nnz} levels={len(ml.levels)} host setup {t_setup:.1f}s")

    t0 = time.time()
    dml = DeviceMultilevelSolver(ml, device=local_rank, graph=not args.no_graph)
    t_upload = time.time() - t0
    if rank == 0:
        for i, (L, dA) in enumerate(zip(ml.levels, dml.A)):
            inf = dA.info()
            log(f"  level {i}: n={L.A.shape[0]} nnz={L.A.nnz} ({L.A.format}) row_ranges={inf['row_blocks']} "
                f"gs_levels fwd/bwd={inf['gs_levels_fwd']}/{inf['gs_levels_bwd']}")
    xd = capi.DeviceArray.from_host(x0)
    bd = capi.DeviceArray.from_host(b)
    stream = dml.stream()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        capi.sync()

    # ---- parity run (also the graph build): first k cycles, residual norms kept
    kpar = 6
    dml.load_device(xd, bd)
    res_gpu = dml.iterate_device(kpar)
    # ---- warmup + timed region: exactly K steps on the resident state
    dml.load_device(xd, bd)
    dml.iterate_device(args.warmup, want_residuals=False)
    barrier()
    e0, e1 = capi.Event(), capi.Event()
    t0 = time.perf_counter()
    e0.record(stream)
    dml.iterate_device(args.steps, want_residuals=False)
    e1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_ms(e1)
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    ms_per_step = wall * 1e3 / args.steps

    # ---- roofline of the dominant bandwidth kernel: fine-level CSR SpMV (r = b - A x),
    #      HIP events on the solver's stream, algorithmic bytes of SURVEY.md 8(d)
    A0 = dml.A[0]
    rd = capi.DeviceArray(n, np.float64)
    reps = 50
    for _ in range(5):
        A0.spmv(capi.SPMV_RESID, xd, rd, b=bd, stream=stream)
    f0, f1 = capi.Event(), capi.Event()
    f0.record(stream)
    for _ in range(reps):
        A0.spmv(capi.SPMV_RESID, xd, rd, b=bd, stream=stream)
    f1.record(stream)
    f1.synchronize()
    spmv_ms = f0.elapsed_ms(f1) / reps
    bytes_resid = spmv_bytes(A) + 8 * n               # + b read
    achieved = bytes_resid / spmv_ms / 1e6            # GB/s
    traffic = None
    pm = ROOT / "profiles" / f"pmc_{args.workload}.json"
    if pm.exists():
        try:
            traffic = json.loads(pm.read_text()).get("resid_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": "csr_stream_kernel<double, RESID> (fine-level r = b - A x)", "bound": "hbm",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "bytes_per_launch": int(bytes_resid), "ms_per_launch": round(spmv_ms, 5)}

    out = None
    if rank == 0:
        # ---- CPU baseline: the reference's own serial solve on this host (1 core), same ml / b
        cpu = None
        res_cpu = None
        if args.cpu_cycles != 0:
            kcpu = args.cpu_cycles if args.cpu_cycles > 0 else (3 if n > 5_000_000 else 10)
            kcpu = max(kcpu, 1)
            r = []
            t0 = time.perf_counter()
            ml.solve(b, x0=x0, tol=1e-30, maxiter=kcpu, residuals=r)
            tcpu = time.perf_counter() - t0
            res_cpu = np.array(r)
            ts = time.perf_counter()
            for _ in range(3):
                A @ b
            t_spmv_cpu = (time.perf_counter() - ts) / 3
            cpu = {"value": round(kcpu / tcpu, 4), "unit": "cycles/s", "cores": 1, "kind": "reference",
                   "sample": f"{kcpu} V-cycles of the same solver/rhs via oracle/_ref (pyamg reference, serial) "
                             f"in {tcpu:.1f}s; fine-level A@x {t_spmv_cpu * 1e3:.1f} ms = "
                             f"{spmv_bytes(A) / t_spmv_cpu / 1e9:.2f} GB/s",
                   "spmv_GBps": round(spmv_bytes(A) / t_spmv_cpu / 1e9, 3)}
        parity = None
        if res_cpu is not None:
            m = min(len(res_cpu) - 1, kpar)
            d = np.abs(res_gpu[:m] - res_cpu[1:m + 1])
            parity = {"cycles_compared": int(m), "max_abs_diff_over_r0": float(np.max(d) / res_cpu[0]),
                      "max_rel_diff": float(np.max(d / res_cpu[1:m + 1])), "tolerance": "1e-10 * ||r0|| (random rhs)"}
        out = {
            # N > 1: order-exact Gauss-Seidel has a global sequential dependency and does not shard in
            # parity mode -> N independent replicas, whole-job value = N x cycles/s
            "metric": "vcycle_iterations_per_sec", "value": round(world * args.steps / wall, 3), "unit": "cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": wl["label"], "key": args.workload, "n": int(n),
                                            "nnz": int(A.nnz), "levels": len(ml.levels),
                                            "cycle": "V(1,1)", "graph": not args.no_graph,
                                            "parallelism": "single GPU" if world == 1 else f"{world} replicas"},
            "event_ms_per_step": round(ev_ms / args.steps, 4),
            "spmv_GBps": round(achieved, 1), "spmv_pct_of_hbm_peak": round(100 * achieved / HBM_PEAK_GBPS, 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            "host": {"setup_s": round(t_setup, 1), "upload_s": round(t_upload, 1), "cores": os.cpu_count()},
            "residuals_gpu": [float(v) for v in res_gpu],
        }
        if cpu:
            out["speedup_vs_cpu_reference"] = round(out["value"] / cpu["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
