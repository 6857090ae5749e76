#!/usr/bin/env python3
"""Headline benchmark: V-cycle iterations/sec + fine-level SpMV GB/s (% of the MI355X HBM
roofline) -- BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c3j|c4s|small] [--no-extras]

A *step* is one pass of the hot path: one multigrid V-cycle on the device-resident hierarchy
followed by the convergence-check residual norm ||b - A x|| (exactly one iteration of the
loop in pyamg/multilevel.py:558-569).  Inputs are resident in HBM when the timed region
starts.  The hierarchy is built on the host by the reference itself (oracle/_ref = the
reference compiled from /root/reference; the setup phase stays on the host, north star) -- by default with
its spectral radii, prolongation smoothing, strength filter and sparse products run through
pyamg_amd.aggregation.device_setup (--host-setup: the reference alone; "host" reports both times) -- and
shipped to HBM once.

N = 1: the workload is BASELINE.json configs[2], the north star's problem: 3-D 7-pt Poisson 256^3,
smoothed aggregation, symmetric Gauss-Seidel V(1,1), fp64.  The line also carries, under "sharded", the
Chebyshev(3) cycle of configs[3] on the same hierarchy run by the resident engine AND by the row-sharded C++ driver
(pamg_dist_*, one rank, no peers: what the driver costs over the resident cycle), and under "extra": configs[1]
(2000^2, weighted Jacobi), configs[4] on one GPU (3-D elasticity, block GS and block Jacobi), configs[0] (the README
anchor) and configs[3] at its own size (512^3 Chebyshev(3) on one GPU; 384^3 if that fails).

N > 1: order-exact Gauss-Seidel has a global sequential dependency and does not shard, so the HEADLINE
of a multi-GPU run is configs[3], the workload the north star shards: 3-D Poisson 512^3, SA, Chebyshev(3), fine levels
row-sharded over the N GPUs (RCCL halo exchange over xGMI overlapped with the interior rows, coarse levels collapsed),
"scaling": "strong" -- the same problem for every N; its N = 1 point is "extra.c4" of the N = 1 line.  The hierarchy is
built ONCE, on rank 0, and shipped to the ranks array by array (DistMultilevelSolver.from_rank0).  N independent
replicas of the Gauss-Seidel workload are reported under "extra.replicas".

Prints ONE JSON line (rank 0): metric/value/unit/... + "roofline" (fine-level CSR SpMV
kernel: algorithmic bytes / HIP-event time vs 8 TB/s) + "cpu_baseline" (the reference's own
serial solve on this host, same hierarchy, same b) + "parity" (residual norms of this run's
GPU cycles vs the reference's cycles).
"""
import argparse
import json
import os
import sys
import time

# the host-side sweep planners first-touch gigabytes from a hundred threads: 2 MB pages where the kernel grants them (csrc/pamg_plan_vec.h; -0.3 s of the
# 256^3 upload on the GPU box, slower inside a small container -- hence an application's choice, not the library's default)
os.environ.setdefault("PAMG_PLAN_HUGEPAGES", "1")
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (~6.3 TB/s achievable)
SEED = 20260924
JAC = ("jacobi", {"omega": 4.0 / 3.0})
GS = ("gauss_seidel", {"sweep": "symmetric"})
CHEB = ("chebyshev", {"degree": 3, "iterations": 1})
WORKLOADS = {
    "c2": dict(grid=(2000, 2000), smoother=JAC,
               label="gallery.poisson((2000,2000)) SA V-cycle, weighted-Jacobi pre/post, fp64"),
    # BASELINE configs[0]: the reference's own README example (README.md:126-153), CPU-runnable; Jacobi pre/post
    "c1": dict(grid=(500, 500), smoother=JAC, kind="rs",
               label="gallery.poisson((500,500)) CSR, ruge_stuben_solver, Jacobi pre/post, fp64"),
    "c3": dict(grid=(256, 256, 256), smoother=GS,
               label="3D 7-pt Poisson 256^3 (16.7M dof) SA V-cycle, symmetric Gauss-Seidel, fp64"),
    "c3j": dict(grid=(256, 256, 256), smoother=JAC,
                label="3D 7-pt Poisson 256^3 SA V-cycle, weighted-Jacobi pre/post, fp64"),
    # BASELINE configs[3] at the largest size whose SERIAL reference setup (aggregation.py:280-431, ~60 s and ~10 GB per
    # 16.7M rows) still fits a bench run: 384^3 = 56.6M rows (the 512^3 setup takes ~16 min and ~78 GB on the host)
    "c4": dict(grid=(384, 384, 384), smoother=CHEB,
               label="3D 7-pt Poisson 384^3 (56.6M dof) SA V-cycle, Chebyshev(3) smoother, fp64"),
    # BASELINE configs[3] at its own size: with the device setup operators the hierarchy is a ~1.5 min build (the reference
    # alone: ~16 min); 134 M rows, ~25 GB resident.  Not part of the default run.
    "c4x": dict(grid=(512, 512, 512), smoother=CHEB,
                label="3D 7-pt Poisson 512^3 (134M dof) SA V-cycle, Chebyshev(3) smoother, fp64"),
    "c4s": dict(grid=(256, 256, 256), smoother=CHEB,
                label="3D 7-pt Poisson 256^3 SA V-cycle, Chebyshev(3) smoother, fp64"),
    # BASELINE configs[4] in miniature: 3-D linear elasticity (P1 tets on an N^3-vertex cube), BSR(3,3),
    # SA with Jacobi prolongation smoothing and 6 rigid-body modes, block Jacobi / the default block GS
    "c5": dict(grid=(64,), elasticity=True, smoother="block_gauss_seidel",
               label="3D linear elasticity 64^3 vertices (BSR 3x3, 774K dof, 6 rigid-body modes) SA with Jacobi prolongation smoothing, V-cycle, block Gauss-Seidel (SA default), fp64"),
    # ... and with the POINT sweep on the block operators (relaxation.gauss_seidel on BSR -> amg_core::bsr_gauss_seidel): fast order through the scalar twin (round 6)
    "c5p": dict(grid=(64,), elasticity=True, smoother=("gauss_seidel", {"sweep": "symmetric"}),
                label="3D linear elasticity 64^3 vertices (BSR 3x3, 774K dof) SA V-cycle, point Gauss-Seidel on the block operators (symmetric), fp64"),
    "c5s": dict(grid=(40,), elasticity=True, smoother="block_jacobi",
                label="3D linear elasticity 40^3 vertices (BSR 3x3, 187K dof) SA V-cycle, block Jacobi, fp64"),
    "c5g": dict(grid=(40,), elasticity=True, smoother="block_gauss_seidel",
                label="3D linear elasticity 40^3 vertices (BSR 3x3, 187K dof) SA V-cycle, block Gauss-Seidel (SA default), fp64"),
    # the "next" rows measured the same way: what pyamg.solve() configures for a NON-symmetric operator
    # (blackbox.py:100-131: energy-minimisation SA + gauss_seidel_nr, 2 symmetric sweeps) and the AIR solver
    # (classical/air.py: FC Jacobi) on upwind convection-diffusion
    "c6n": dict(grid=(384, 384), convdiff=3.0, kind="blackbox", smoother="gauss_seidel_nr",
                label="2D upwind convection-diffusion 384^2, pyamg.solve() configuration (SA energy-min, gauss_seidel_nr x2 symmetric) V-cycle, fp64"),
    "c6n3": dict(grid=(64, 64, 64), convdiff=3.0, kind="blackbox", smoother="gauss_seidel_nr",
                 label="3D upwind convection-diffusion 64^3, pyamg.solve() configuration (SA energy-min, gauss_seidel_nr x2 symmetric) V-cycle, fp64"),
    "c7a": dict(grid=(512, 512), convdiff=3.0, kind="air", smoother="fc_jacobi",
                label="2D upwind convection-diffusion 512^2, AIR V-cycle (FC Jacobi: 2 F-sweeps + 1 C-sweep), fp64"),
    # the "next" row f2's last smoother: multiplicative overlapping Schwarz (relaxation.py:157-262), one subdomain per row (its pattern) -- 4 n dependency
    # levels on an n x n grid; since round 6 one persistent launch per sweep (csrc/pamg_schwarz.hip)
    "c8s": dict(grid=(384, 384), smoother=("schwarz", {"sweep": "symmetric"}),
                label="2D 5-pt Poisson 384^2 SA V-cycle, overlapping Schwarz (symmetric sweeps, one subdomain per row), fp64"),
    "small": dict(grid=(64, 64, 64), smoother=GS,
                  label="3D 7-pt Poisson 64^3 SA V-cycle, symmetric Gauss-Seidel, fp64 (smoke)"),
}


def cpu_quota_cores():
    """cores the container's cgroup grants (cpu.max), None without a quota -- os.cpu_count() reports the machine (the pool's GPU boxes: 256 threads, quota 16)"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 1)
    except Exception:                                   # noqa: BLE001
        return None


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


# ---- what goes to stdout.  Round 5's line had grown to 20.8 KB and the driver no longer parsed it (BENCH_r05.json: parsed = null).  The
#      full record of a run now goes to a FILE (gpurun_out/bench_detail.json, a copy under profiles/ is committed per round); stdout carries
#      ONE compact line of flat scalars -- the contract's keys, `roofline`, `cpu_baseline` and one scalar per extra leg.
HEADLINE_LIMIT = 8_000          # bytes; tests/test_host.py::test_bench_headline_stays_small builds a full fake record against it
DETAIL_PATH = ROOT / "gpurun_out" / "bench_detail.json"


def _short(v, n=160):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "…"


def _flat(d, keep_str=(), n=160):
    """scalars of a dict (numbers, booleans, None) + the named strings, shortened"""
    out = {}
    for k, v in (d or {}).items():
        if isinstance(v, bool) or v is None or isinstance(v, (int, float)):
            out[k] = v
        elif isinstance(v, str) and k in keep_str:
            out[k] = _short(v, n)
    return out


def headline(out):
    """the compact stdout line of a run: flat scalars only, <= HEADLINE_LIMIT bytes whatever the legs put into `out`"""
    h = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    h["config"] = _flat(out.get("config"), keep_str=("workload", "key", "cycle", "gs_order", "parallelism", "n1_point", "modelled_note"), n=200)
    for k in ("event_ms_per_step", "spmv_GBps", "spmv_pct_of_hbm_peak", "speedup_vs_cpu_reference", "error"):
        if k in out:
            h[k] = _short(out[k], 300)
    rf = out.get("roofline")
    h["roofline"] = _flat(rf, keep_str=("kernel", "bound", "unit", "structured_kernel", "dominant_kernel"), n=140) if rf else None
    cpu = out.get("cpu_baseline")
    h["cpu_baseline"] = _flat(cpu, keep_str=("unit", "kind", "sample"), n=200) if cpu else None
    par = out.get("parity") or {}
    if par:
        h["parity"] = _flat(par)
        rp = par.get("reference_protocol") or {}
        if rp:
            h["parity"].update({"protocol_ok": rp.get("ok"), "protocol_max_rel_diff": rp.get("max_rel_diff"), "protocol_cycles": rp.get("cycles_compared")})
        sv = par.get("sharded_vs_resident") or {}
        if sv:
            h["parity"].update({"sharded_vs_resident_ok": sv.get("ok"), "sharded_vs_resident_max_rel_diff": sv.get("max_rel_diff")})
    if out.get("host"):
        h["host"] = _flat(out["host"])
    ac = out.get("accel") or {}
    for k, v in ac.items():
        if k.startswith("hl_"):
            h["accel_" + k[3:]] = v
    t = out.get("time_to_tol_1e-8") or {}
    if t:
        h["cycles_to_tol_1e-8"], h["seconds_to_tol_1e-8"] = t.get("cycles"), t.get("seconds")
    for r_ in (out.get("gs_sweeps") or {}).get("per_level", []):
        h[f"gs_sweep_ms_level{r_['level']}"] = r_.get("ms_per_forward_sweep")
        h[f"gs_sweep_dependency_levels_level{r_['level']}"] = r_.get("dependency_levels")
        if r_.get("hand_offs") is not None:
            h[f"gs_sweep_hand_offs_level{r_['level']}"] = r_.get("hand_offs")
    if isinstance(out.get("exact_order"), dict):
        h["exact_order_ms_per_step"] = out["exact_order"].get("ms_per_step")
    sh = out.get("sharded") or {}
    if sh:
        h["cheby_resident_ms_per_step"] = sh.get("ms_per_step")
        h["cheby_sharded_driver_one_rank_ms_per_step"] = (sh.get("sharded_driver_one_rank") or {}).get("ms_per_step")
    sd = out.get("sharded_driver") or {}
    if sd:
        h["sharded_driver"] = _flat(sd, keep_str=("transport",))
    # one group of scalars per extra leg
    for key, leg in (out.get("extra") or {}).items():
        if not isinstance(leg, dict):
            continue
        legs = [(f"extra_{key}", leg)] + [(f"extra_{key}_{k}", v) for k, v in leg.items() if isinstance(v, dict) and "ms_per_step" in v]
        for tag, lg in legs:
            if "error" in lg:
                h[tag + "_error"] = _short(str(lg["error"]), 120)
            if "ms_per_step" in lg:
                h[tag + "_ms"], h[tag + "_cycles_per_s"] = lg.get("ms_per_step"), lg.get("value")
            cb = lg.get("cpu_baseline") or {}
            if cb:
                h[tag + "_cpu_cycles_per_s"] = cb.get("value")
            rp = (lg.get("parity") or {}).get("reference_protocol") or {}
            if rp:
                h[tag + "_protocol_ok"], h[tag + "_protocol_max_rel_diff"] = rp.get("ok"), rp.get("max_rel_diff")
            tt = lg.get("time_to_tol_1e-8") or {}
            if tt.get("cycles") is not None:
                h[tag + "_cycles_to_tol"], h[tag + "_s_to_tol"] = tt.get("cycles"), tt.get("seconds")
            for k, v in lg.items():                       # scalars a leg marks for the line (accelerated solves, ...)
                if k.startswith("hl_") and (isinstance(v, (int, float, bool)) or v is None):
                    h[tag + "_" + k[3:]] = v
    ms_ = out.get("modelled_scaling") or {}
    for r_ in ms_.get("rows", []):
        h[f"modelled_ms_n{r_['n']}"] = r_.get("ms_per_step")
        if "slowest_rank" in r_:
            h[f"modelled_slowest_rank_n{r_['n']}"] = r_["slowest_rank"]
    if ms_.get("rows"):
        h["modelled_ms_n1_measured"] = ms_.get("n1_ms_per_step_measured")
    if out.get("notes"):
        h["notes"] = [_short(str(s), 120) for s in out["notes"]][:3]
    h["detail"] = "gpurun_out/bench_detail.json (the full record of this run; committed copies: profiles/rNN_bench_detail*.json)"
    # the line must parse whatever happens: shed the least important keys until it fits
    for drop in ("notes", "host", "sharded_driver", "parity"):
        if len(json.dumps(h)) <= HEADLINE_LIMIT:
            break
        h.pop(drop, None)
    if len(json.dumps(h)) > HEADLINE_LIMIT:
        for k in [k for k in h if k.startswith(("extra_", "gs_sweep_", "modelled_"))][::-1]:
            h.pop(k)
            if len(json.dumps(h)) <= HEADLINE_LIMIT:
                break
    return h


def write_detail(out, path=None):
    path = Path(path or os.environ.get("PAMG_BENCH_DETAIL", DETAIL_PATH))
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(json.dumps(out, indent=1))
        return str(path)
    except OSError as e:                                        # a read-only tree must not cost the line
        log(f"detail record not written: {e!r}")
        return None


def measure_traffic(grid, nrows, general=False):
    """HBM bytes per launch of the fine-level residual kernel, counted in THIS run: tools/spmv_pmc.py (the same operator,
    the same kernel) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only),
    corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 128-byte requests at
    64 bytes: x 2; WRITE_SIZE 1:1; both in KiB).  None if the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3") or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None
    vals = {}
    names = ("csr_stream_kernel",) if general else ("csr_stream_kernel", "csr_rowgather_kernel", "csr_rowpat_kernel", "csr_rowmask")
    tmp = tempfile.mkdtemp(prefix="pamg_pmc_")
    try:
        for cname in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, cname)
            cmd = ["rocprofv3", "--pmc", cname, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, str(ROOT / "tools" / "spmv_pmc.py")] + [str(g) for g in grid] + (["--general=1"] if general else [])
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True,
                           env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
            tot, cnt = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") == cname and any(k in r.get("Kernel_Name", "") for k in names) and \
                                int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) >= 256 * 1024:
                            tot += float(r["Counter_Value"])
                            cnt += 1
            if cnt == 0:
                return None
            vals[cname] = 1024.0 * tot / cnt
        fetch = 2.0 * vals["FETCH_SIZE"]
        return {"bytes_per_launch": int(fetch + vals["WRITE_SIZE"]), "fetch_size_raw": int(vals["FETCH_SIZE"]),
                "fetch_corrected_x2": int(fetch), "write_size": int(vals["WRITE_SIZE"]),
                "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE on tools/spmv_pmc.py, this run"}
    except Exception as e:                                      # noqa: BLE001 -- a missing profiler must not fail the bench
        log(f"PMC pass skipped: {e!r}")
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, the same
    environment contract torch.distributed.run provides), pass rank 0's JSON line through, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"bench ranks exited with {rcs}")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("PAMG_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-cycles", type=int, default=-1, help="reference cycles timed on the host (-1 auto, 0 skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the sharded-Chebyshev and configs[1] legs")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-model", action="store_true", help="skip the modelled 2 / 4 / 8 GPU curve of the 512^3 Chebyshev workload (SURVEY 8e availability caveat)")
    ap.add_argument("--model-ranks", default="all", choices=("all", "mid"), help="the modelled N-GPU curve times every rank of a partition (max over ranks) or rank N // 2 only")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes that measure the SpMV's HBM traffic")
    ap.add_argument("--min-rows", type=int, default=200_000, help="shard levels with at least this many rows")
    ap.add_argument("--host-setup", action="store_true", help="build the hierarchies with the reference alone (no device setup operators)")
    ap.add_argument("--no-setup-compare", action="store_true", help="skip the second, reference-only setup of the main workload")
    ap.add_argument("--protocol-cycles", type=int, default=10, help="cycles of the reference-protocol parity run (b = 0, x0 = rand); 0 skips it")
    ap.add_argument("--kernel-map", default=None, help="write the launch -> (level, operator, bytes) map of the main workload's cycle here (tools/summarize_prof.py joins it with a rocprofv3 trace)")
    ap.add_argument("--shard-workload", default=os.environ.get("PAMG_SHARD_WORKLOAD", "c4x"), choices=sorted(WORKLOADS),
                    help="N > 1: the row-sharded headline workload (BASELINE configs[3] = c4x, 512^3 Chebyshev)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # torch is plumbing only (process group, the contract's cuda synchronize, comm buffers of the
    # sharded path).  It must be imported BEFORE libpyamg_amd.so is loaded: both link
    # libamdhip64.so.7 and torch's copy has to be the one the process binds (measured on the
    # MI355X box: the other order leaves torch without devices).
    import torch
    import numpy as np
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PAMG_BENCH_BACKEND=gloo + PAMG_BENCH_ONE_GPU=1: rehearsal of the N > 1 control flow on a
        # single-GPU box (all ranks on device 0, host-staged transport); production = nccl (RCCL)
        backend = os.environ.get("PAMG_BENCH_BACKEND", "nccl")
        if os.environ.get("PAMG_BENCH_ONE_GPU"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        from datetime import timedelta
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timedelta(minutes=45))
        else:
            dist.init_process_group(backend, timeout=timedelta(minutes=45))
    from pyamg_amd import DeviceMultilevelSolver, _capi as capi
    from pyamg_amd.hierarchy import extract
    from tools.problems import spmv_bytes
    capi.check(capi.lib().pamg_set_device(local_rank), "set_device")

    import oracle.refimport as ri
    if not ri.available():
        raise SystemExit("bench.py needs the reference build oracle/_ref for the host-side setup phase "
                         "(python oracle/build_ref.py in the build container)")
    import pyamg
    from pyamg.relaxation.smoothing import change_smoothers

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        capi.sync()

    def max_over_ranks(v):
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    import contextlib
    from pyamg_amd.aggregation import device_setup

    def setup_ctx(device, prolongation=True):
        # the reference's setup with its spectral radii / prolongation smoothing run by pyamg_amd.aggregation on the GPU
        return device_setup(pyamg, prolongation=prolongation) if device else contextlib.nullcontext()

    assembly_s = [0.0]

    def build(wl, device=None):
        # returns (A, ml, seconds of the HIERARCHY setup); assembling the test operator (scipy kron products: ~1.4 s at 256^3) is timed apart
        # (assembly_s[0]) -- it is the caller's problem, not the solver's setup, and rounds 2-5 had it inside `setup_s`
        device = (not args.host_setup) if device is None else device
        t0 = time.time()
        if wl.get("elasticity"):
            from tools.problems import elasticity3d
            A, B = elasticity3d(wl["grid"][0])
            assembly_s[0] = time.time() - t0
            t0 = time.time()
            np.random.seed(SEED)
            with setup_ctx(device):                          # block operators included: bsr_matmat / bsr_binop_bsr semantics on the device
                ml = pyamg.smoothed_aggregation_solver(A, B=B, smooth="jacobi", presmoother=wl["smoother"],
                                                       postsmoother=wl["smoother"], max_coarse=10)
            return A, ml, time.time() - t0
        if wl.get("convdiff"):
            import scipy.sparse as sp
            mx, my = wl["grid"][:2]
            Dx = sp.diags_array([np.ones(mx), -np.ones(mx - 1)], offsets=[0, -1], shape=(mx, mx))
            Dy = sp.diags_array([2 * np.ones(my), -np.ones(my - 1), -np.ones(my - 1)], offsets=[0, -1, 1], shape=(my, my))
            A = sp.csr_array(wl["convdiff"] * sp.kron(sp.eye_array(my), Dx) + sp.kron(Dy, sp.eye_array(mx)))
            if len(wl["grid"]) == 3:                     # + diffusion in z
                mz = wl["grid"][2]
                Dz = sp.diags_array([2 * np.ones(mz), -np.ones(mz - 1), -np.ones(mz - 1)], offsets=[0, -1, 1], shape=(mz, mz))
                A = sp.csr_array(sp.kron(sp.eye_array(mz), A) + sp.kron(Dz, sp.eye_array(mx * my)))
            A.sort_indices()
            assembly_s[0] = time.time() - t0
            t0 = time.time()
            np.random.seed(SEED)
            if wl["kind"] == "air":
                ml = pyamg.air_solver(A, max_coarse=20)
            else:
                from pyamg import blackbox
                ml = blackbox.solver(A, blackbox.solver_configuration(A, verb=False))
            return A, ml, time.time() - t0
        A = pyamg.gallery.poisson(wl["grid"], format="csr")
        assembly_s[0] = time.time() - t0
        t0 = time.time()
        np.random.seed(SEED)                   # Arnoldi start vectors of the smoother setup
        with setup_ctx(device):
            if wl.get("kind") == "rs":
                ml = pyamg.ruge_stuben_solver(A, presmoother=wl["smoother"], postsmoother=wl["smoother"])
            else:
                ml = pyamg.smoothed_aggregation_solver(A, presmoother=wl["smoother"], postsmoother=wl["smoother"],
                                                       max_coarse=10)
        return A, ml, time.time() - t0

    def rhs(n):
        np.random.seed(SEED)
        return np.random.rand(n), np.zeros(n)

    def time_resident(dml, b, x0, steps, warmup, kpar=6):
        """exactly `steps` steps on the resident state, bracketed by barrier + synchronize"""
        xd, bd = capi.DeviceArray.from_host(x0), capi.DeviceArray.from_host(b)
        dml.load_device(xd, bd)
        res = dml.iterate_device(kpar)                    # parity run (also builds the graph)
        dml.load_device(xd, bd)
        dml.iterate_device(warmup, want_residuals=False)
        barrier()
        e0, e1 = capi.Event(), capi.Event()
        stream = dml.stream()
        t0 = time.perf_counter()
        e0.record(stream)
        dml.iterate_device(steps, want_residuals=False)
        e1.record(stream)
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        return wall, e0.elapsed_ms(e1), res, xd, bd

    def spmv_cpu_ms(A, v, reps=7):
        """SciPy's serial A @ v on this host: two untimed products (page-in, allocator), then the median of `reps`"""
        for _ in range(2):
            A @ v
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            A @ v
            ts.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(ts))

    def cpu_reference(ml, A, b, x0, kcpu):
        r = []
        t0 = time.perf_counter()
        ml.solve(b, x0=x0, tol=1e-30, maxiter=kcpu, residuals=r)
        tcpu = time.perf_counter() - t0
        tsp = spmv_cpu_ms(A, b)
        return {"value": round(kcpu / tcpu, 4), "unit": "cycles/s", "cores": 1, "kind": "reference",
                "sample": f"{kcpu} V-cycles of the same solver/rhs by the reference (oracle/_ref, serial) in {tcpu:.1f}s; "
                          f"fine-level A@x {tsp:.1f} ms (median of 7 after 2 warm-up products)",
                "spmv_ms": round(tsp, 3)}, np.array(r)

    def parity_of(res_gpu, res_cpu):
        m = min(len(res_cpu) - 1, len(res_gpu))
        d = np.abs(np.asarray(res_gpu)[:m] - res_cpu[1:m + 1])
        return {"cycles_compared": int(m), "max_abs_diff_over_r0": float(np.max(d) / res_cpu[0]),
                "max_rel_diff": float(np.max(d / res_cpu[1:m + 1])), "tolerance": "1e-10 * ||r0|| (random rhs)"}

    def protocol_parity(dml, ml, n, k=10):
        """the reference's own convergence protocol (docs/paper/example.py:11-14): b = 0, x0 = rand -- r = -A x has no
        cancellation, so every residual norm of the first k cycles must agree to 1e-10 RELATIVE"""
        np.random.seed(2022)
        x0r = np.random.rand(n)
        b0 = np.zeros(n)
        xz, bz = capi.DeviceArray.from_host(x0r), capi.DeviceArray.from_host(b0)
        dml.load_device(xz, bz)
        g = np.asarray(dml.iterate_device(k))
        xz.free(); bz.free()
        r = []
        t0 = time.perf_counter()
        ml.solve(b0, x0=x0r, tol=1e-30, maxiter=k, residuals=r)
        tc = time.perf_counter() - t0
        c = np.asarray(r)[1:k + 1]
        m = min(len(g), len(c))
        rel = np.abs(g[:m] - c[:m]) / c[:m]
        return {"protocol": "b = 0, x0 = rand (seed 2022), reference residual norms after each cycle", "cycles_compared": int(m),
                "max_rel_diff": float(rel.max()), "tolerance": "1e-10 relative, every cycle", "ok": bool(rel.max() <= 1e-10),
                "first_last_norm": [float(c[0]), float(c[m - 1])], "cpu_s": round(tc, 1), "cpu_cycles": int(k)}

    def time_to_tol(dml_, b_, x0_, tol=1e-8, maxit=400):
        """multilevel.py:558-580 with the reference's stopping rule ||b - A x|| < tol * ||b||: the number of cycles it takes
        from x0 and the wall time of exactly those cycles (each followed by its convergence-check norm) on the resident state"""
        xd_, bd_ = capi.DeviceArray.from_host(x0_), capi.DeviceArray.from_host(b_)
        normb = float(np.linalg.norm(b_)) or 1.0
        dml_.load_device(xd_, bd_)
        res_, k = [], None
        while len(res_) < maxit and k is None:
            res_ += list(dml_.iterate_device(20))
            hit = [i for i, r in enumerate(res_) if r < tol * normb]
            k = hit[0] + 1 if hit else None
        out_ = {"tol": tol, "cycles": k, "rule": "||b - A x|| < tol * ||b|| (multilevel.py:571)", "final_relative_residual": float(res_[(k or len(res_)) - 1] / normb)}
        if k is not None:
            dml_.load_device(xd_, bd_)
            barrier()
            t0_ = time.perf_counter()
            dml_.iterate_device(k, want_residuals=False)
            barrier()
            out_["seconds"] = round(time.perf_counter() - t0_, 5)
        xd_.free(); bd_.free()
        return out_

    def accel_leg(dml_, ml_, b_, x0_, accel, tol=1e-8, maxiter=200, ref_iters=4):
        """f1: the way the reference is used (multilevel.py:479-535, blackbox.py:285-311) -- a Krylov method preconditioned by one cycle.  Device:
        solve(b, tol, accel=...) with the Krylov loop resident (pamg_solver_pcg / gmres), host vectors in and out (the PCIe copies are inside
        the time); the second of two runs is timed.  Reference: the same call with maxiter = ref_iters on the same hierarchy, histories compared."""
        out_ = {"accel": accel, "tol": tol}
        r1 = []
        dml_.solve(b_, x0=x0_, tol=tol, maxiter=maxiter, accel=accel, residuals=r1)              # warm-up (graphs, work vectors)
        r2 = []
        t0_ = time.perf_counter()
        dml_.solve(b_, x0=x0_, tol=tol, maxiter=maxiter, accel=accel, residuals=r2)
        out_["seconds_to_tol"] = round(time.perf_counter() - t0_, 5)
        out_["iterations"] = len(r2) - 1
        out_["final_residual_over_first"] = float(r2[-1] / r2[0]) if r2 and r2[0] else None
        out_["iterations_per_s"] = round((len(r2) - 1) / max(out_["seconds_to_tol"], 1e-9), 2)
        if ref_iters > 0:
            rr = []
            t0_ = time.perf_counter()
            ml_.solve(b_, x0=x0_, tol=tol, maxiter=ref_iters, accel=accel, residuals=rr)
            tc_ = time.perf_counter() - t0_
            m_ = min(len(rr), len(r2))
            d_ = np.abs(np.asarray(rr[:m_]) - np.asarray(r2[:m_]))
            out_["reference"] = {"iterations_timed": len(rr) - 1, "seconds": round(tc_, 2), "seconds_per_iteration": round(tc_ / max(1, len(rr) - 1), 3),
                                 "norms_compared": int(m_), "max_abs_diff_over_r0": float(d_.max() / rr[0]), "tolerance": "1e-10 * ||r0||",
                                 "ok": bool(d_.max() <= 1e-10 * rr[0])}
            out_["cpu_seconds_to_tol_extrapolated"] = round(out_["reference"]["seconds_per_iteration"] * out_["iterations"], 1)
        # scalars for the headline
        out_["hl_" + accel + "_s_to_tol"] = out_["seconds_to_tol"]
        out_["hl_" + accel + "_iterations"] = out_["iterations"]
        if "reference" in out_:
            out_["hl_" + accel + "_parity_ok"] = out_["reference"]["ok"]
            out_["hl_" + accel + "_cpu_s_per_iteration"] = out_["reference"]["seconds_per_iteration"]
        return out_

    def cpu_from_protocol(pp, A, v):
        """the cpu_baseline of a leg whose only reference run is the protocol run (512^3: one reference cycle is ~45 s)"""
        tsp = spmv_cpu_ms(A, v, reps=5)
        return {"value": round(pp["cpu_cycles"] / pp["cpu_s"], 4), "unit": "cycles/s", "cores": 1, "kind": "reference",
                "sample": f"{pp['cpu_cycles']} V-cycles of the same solver by the reference (oracle/_ref, serial) in {pp['cpu_s']:.1f}s "
                          f"(the b = 0 / x0 = rand protocol run); fine-level A@x {tsp:.1f} ms (median of 5 after 2 warm-up products)",
                "spmv_ms": round(tsp, 3)}

    SETUP_NOTE = ("reference (oracle/_ref) alone" if args.host_setup else
                  "reference (oracle/_ref) with pyamg_amd.aggregation.device_setup: spectral radii (Arnoldi), strength filter, "
                  "prolongation smoothing and the Galerkin products on the GPU")

    import threading
    printed = threading.Event()
    box = {"out": None}

    # ---- the modelled 1 -> 8 GPU curve (SURVEY 8e "availability caveat": no multi-GPU node => the ranks' work on one device + modelled xGMI
    #      time).  A MODEL, labelled as such everywhere it appears.  For N ranks the hierarchy is partitioned exactly as the N > 1 run does
    #      (pyamg_amd.dist.ShardedHierarchy); the INTERIOR rank N // 2 (two neighbours: the critical one) runs its own launches on this GPU
    #      through the C++ driver with nobody on the wire (pamg_dist_set_model_transport): pack, interior ranges, boundary ranges, collapse,
    #      replicated tail -- the compute critical path.  The wire is added from that rank's exchange plan:
    #        per halo exchange    t_x = XGMI_MSG_LATENCY_US + (largest message from one neighbour) / XGMI_LINK_GBPS   (every neighbour on its own link)
    #        exposed per exchange max(0, t_x - interior launch of that level's operator)  [overlap assumed]  or  t_x  [no overlap]
    #        all-reduces          (collapse: the coarse right-hand side; norm: one value)  ALLREDUCE_LATENCY_US + 2 (N - 1) / N * bytes / XGMI_LINK_GBPS
    XGMI_LINK_GBPS = 153.0              # per link, the figure this round's hardware notes give (7 links per GPU, point to point)
    XGMI_MSG_LATENCY_US = 12.0          # one grouped ncclSend / ncclRecv batch end to end (launch on the comm stream + handshake): ASSUMPTION
    ALLREDUCE_LATENCY_US = 25.0         # small-message ring all-reduce over 8 GPUs: ASSUMPTION

    def model_scaling(spec, label, n1_ms, ranks=(2, 4, 8), cycles=8):
        from pyamg_amd.dist import DeviceOps, DistMultilevelSolver, ShardedHierarchy
        rows = []
        for N in ranks:
            # EVERY rank of the N-rank partition is timed (round 5 timed rank N // 2 alone and assumed it the slowest): the modelled step is the
            # MAX over ranks of compute + exposed wire, as the real run's barrier makes it
            shared_ = {}
            per_rank = []
            for r in (range(N) if args.model_ranks == "all" else [N // 2]):
                row_ = model_one_rank(spec, label, n1_ms, N, r, shared_, cycles)
                per_rank.append(row_)
            worst = max(per_rank, key=lambda q: q["ms_per_step"])
            row = dict(worst)
            row["slowest_rank"] = worst["rank_timed"]
            row["ranks_timed"] = len(per_rank)
            row["per_rank_ms_per_step"] = [q["ms_per_step"] for q in per_rank]
            row["per_rank_compute_ms"] = [q["compute_ms"] for q in per_rank]
            row["per_rank_halo_rows_level1"] = [q["levels"][1]["halo"] if len(q["levels"]) > 1 else None for q in per_rank]
            rows.append(row)
            log(f"modelled scaling {label} N={N}: max over {len(per_rank)} ranks = rank {row['slowest_rank']}: {row['ms_per_step']:.3f} ms per cycle "
                f"({row['speedup_vs_n1']}x of N = 1's {n1_ms:.3f}); per rank {row['per_rank_ms_per_step']}")
        return {"what": "MODEL, not a measurement: every rank's launches timed on this one GPU through the C++ sharded driver with nobody on the wire, "
                        "plus the halo exchanges of its plan at the stated xGMI figures; the step is the MAX over ranks (bench.py model_scaling; DESIGN 6)",
                "workload": label, "n1_ms_per_step_measured": round(n1_ms, 4),
                "assumptions": {"xgmi_link_GBps": XGMI_LINK_GBPS, "message_latency_us": XGMI_MSG_LATENCY_US, "allreduce_latency_us": ALLREDUCE_LATENCY_US,
                                "neighbours_on_separate_links": True, "compute": "stream-ordered launches (the production default with RCCL; one hipGraph is the opt-in PAMG_DIST_GRAPH=1)"},
                "rows": rows}

    def model_one_rank(spec, label, n1_ms, N, r, shared_, cycles):
        from pyamg_amd.dist import DeviceOps, DistMultilevelSolver, ShardedHierarchy
        rows = []
        if True:
            t0m = time.time()
            part = ShardedHierarchy(spec, r, N, args.min_rows, _shared=shared_)
            part.detach()
            sol = DistMultilevelSolver.model_rank(part, DeviceOps(local_rank, spec.dtype))
            nat = sol.native
            n_own = part.plans[0].n_owned_s
            rng_ = np.random.RandomState(7)
            ms = {}
            for tag, use_graph in (("stream_ordered", 0), ("one_graph", 1)):
                nat.set_options(use_graph=use_graph)
                nat.load(rng_.rand(n_own), rng_.rand(n_own))
                nat.iterate(3, want_residuals=False)
                t1 = time.perf_counter()
                nat.iterate(cycles, want_residuals=False)
                ms[tag] = (time.perf_counter() - t1) * 1e3 / cycles
            # interior launch of every sharded level's operator (what an exchange can hide behind), and the exchange plan
            lv, wire_ov, wire_no = [], 0.0, 0.0
            for l in range(part.ns + 1):
                li = nat.level_info(l)
                t_int = 0.0
                if l < part.ns and li["exchanges_per_iteration"]:
                    nl_ = part.plans[l].n_local_s
                    xv, yv = capi.DeviceArray.from_host(rng_.rand(nl_)), capi.DeviceArray(part.plans[l].n_owned_s, np.float64)
                    try:
                        for _ in range(3):
                            sol.A[l].spmv(capi.SPMV_SET, xv, yv, part=1)
                        q0, q1 = capi.Event(), capi.Event()
                        q0.record()
                        for _ in range(10):
                            sol.A[l].spmv(capi.SPMV_SET, xv, yv, part=1)
                        q1.record(); q1.synchronize()
                        t_int = q0.elapsed_ms(q1) / 10
                    finally:
                        xv.free(); yv.free()
                msg_bytes = 8 * max(li["max_received_from_one_peer"], li["max_sent_to_one_peer"])
                t_x = (XGMI_MSG_LATENCY_US * 1e-3 + msg_bytes / (XGMI_LINK_GBPS * 1e6)) if li["exchanges_per_iteration"] and msg_bytes else 0.0
                wire_no += li["exchanges_per_iteration"] * t_x
                wire_ov += li["exchanges_per_iteration"] * max(0.0, t_x - t_int)
                lv.append({"level": l, "owned": li["owned"], "halo": li["halo"], "exchanges_per_iteration": li["exchanges_per_iteration"],
                           "largest_message_bytes": int(msg_bytes), "exchange_ms": round(t_x, 5), "interior_launch_ms": round(t_int, 5)})
            ar = 2 * (ALLREDUCE_LATENCY_US * 1e-3) + 2 * (N - 1) / N * (8 * part.nc + 8) / (XGMI_LINK_GBPS * 1e6)
            rows.append({"n": N, "model": True, "rank_timed": r, "sharded_levels": part.ns,
                         "compute_ms": round(ms["stream_ordered"], 4), "compute_ms_one_graph": round(ms["one_graph"], 4),
                         "exchange_ms_not_overlapped": round(wire_no, 4), "exchange_ms_exposed_with_overlap": round(wire_ov, 4), "allreduce_ms": round(ar, 4),
                         "ms_per_step": round(ms["stream_ordered"] + wire_ov + ar, 4), "ms_per_step_no_overlap": round(ms["stream_ordered"] + wire_no + ar, 4),
                         "overlap_assumed": True, "speedup_vs_n1": round(n1_ms / (ms["stream_ordered"] + wire_ov + ar), 2),
                         "efficiency": round(n1_ms / (ms["stream_ordered"] + wire_ov + ar) / N, 3), "levels": lv, "model_build_s": round(time.time() - t0m, 1)})
            log(f"  rank {r} of {N}: compute {ms['stream_ordered']:.3f} ms (one graph {ms['one_graph']:.3f}), wire exposed {wire_ov:.3f} / all {wire_no:.3f}, "
                f"all-reduces {ar:.3f} -> {rows[-1]['ms_per_step']:.3f} ms per cycle")
            nat.free()
            for m_ in sol.A + sol.P + sol.R:
                m_.free()
            sol.coarse.free()
            del sol, nat, part
        return rows[0]

    def emit():
        if rank == 0 and box["out"] is not None and not printed.is_set():
            printed.set()
            write_detail(box["out"])                       # the full record: a file, not stdout
            print(json.dumps(headline(box["out"])), flush=True)

    def start_watchdog(budget, what):
        """whatever happens in the legs that follow (an exception on one rank, a collective that never completes on a node
        this code has not run on yet): the line assembled so far is printed exactly once and the process ends"""
        def bail():
            if rank == 0 and box["out"] is not None:
                box["out"].setdefault("notes", []).append(f"{what} exceeded {budget:.0f} s; reported without them")
            emit()
            os._exit(0)
        t = threading.Timer(budget, bail)
        t.daemon = True
        t.start()
        return t

    # =========================================================== N > 1: the row-sharded workload is the headline
    if world > 1:
        from pyamg_amd.dist import DeviceOps, DistMultilevelSolver
        wls = WORKLOADS[args.shard_workload]

        def fail_line(why):
            # a multi-GPU run that cannot produce its number still prints ONE line saying so
            if rank == 0 and not printed.is_set():
                printed.set()
                print(json.dumps({"metric": "vcycle_iterations_per_sec", "value": None, "unit": "cycles/s", "n_gpus": world,
                                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                                  "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                                  "config": {"workload": wls["label"], "key": args.shard_workload}, "error": why}), flush=True)
            os._exit(1)

        budget0 = float(os.environ.get("PAMG_SHARD_TIMEOUT", "1500"))
        wd0 = threading.Timer(budget0, lambda: fail_line(f"the sharded run did not finish within {budget0:.0f} s"))
        wd0.daemon = True
        wd0.start()
        import traceback
        _excepthook = sys.excepthook

        def hook(tp, val, tb):
            traceback.print_exception(tp, val, tb)
            fail_line(f"rank {rank}: {val!r}"[:400])
        sys.excepthook = hook
        if wls["smoother"] == GS or isinstance(wls["smoother"], str):
            raise SystemExit("--shard-workload needs a row-independent smoother (Jacobi / Chebyshev)")
        t0 = time.time()
        spec, n, nnz, nlev, t_setup = None, 0, 0, 0, 0.0
        if rank == 0:
            A, ml, t_setup = build(wls)
            n, nnz, nlev = int(A.shape[0]), int(A.nnz), len(ml.levels)
            spec = extract(ml)
            log(f"sharded workload {args.shard_workload}: n={n} nnz={nnz} levels={nlev} host setup {t_setup:.1f}s (rank 0 only)")
        meta = [n, nnz, nlev]
        dist.broadcast_object_list(meta, src=0)
        n, nnz, nlev = meta
        t1 = time.time()
        d2 = DistMultilevelSolver.from_rank0(spec, ops=DeviceOps(local_rank, np.float64), min_rows=args.min_rows)
        t_ship = time.time() - t1
        b, x0 = rhs(n)
        kpar = 4
        d2.load(b, x0)
        res_sh = d2.iterate(kpar)                                  # parity run (and graph / communicator warm-up)
        d2.load(b, x0)
        d2.iterate(args.warmup, want_residuals=False)
        barrier()
        t0 = time.perf_counter()
        d2.iterate(args.steps)                                     # every step ends with the all-reduced convergence-check norm
        barrier()
        wall = max_over_ranks(time.perf_counter() - t0)
        info = d2.native.info() if d2.native is not None else {}
        # fine-level residual kernel on this rank's shard (interior + boundary ranges = the whole shard), HIP events
        roofline = None
        if rank == 0:
            A0, p0 = d2.A[0], d2.sh.plans[0]
            op0 = d2.sh.A[0]
            by = 12 * op0.nnz + 4 * (p0.n_owned_s + 1) + 8 * p0.n_local_s + 8 * p0.n_owned_s + 8 * p0.n_owned_s
            xs_, bs_, rs_ = (capi.DeviceArray(p0.n_local_s, np.float64) for _ in range(3))
            xs_.upload(np.random.RandomState(0).rand(p0.n_local_s))
            bs_.upload(np.random.RandomState(1).rand(p0.n_local_s))
            for _ in range(3):
                A0.spmv(capi.SPMV_RESID, xs_, rs_, b=bs_)
            f0, f1 = capi.Event(), capi.Event()
            f0.record(None)
            for _ in range(20):
                A0.spmv(capi.SPMV_RESID, xs_, rs_, b=bs_)
            f1.record(None)
            f1.synchronize()
            ms = f0.elapsed_ms(f1) / 20
            # frac on the bytes the running format must move (a compressed operator stream does not move the CSR formula's bytes)
            npat_, nval_ = A0.row_patterns(), A0.value_codes()
            masks_ = A0.row_masks() if npat_ else {"entries": 0}
            vecs_ = 8 * p0.n_local_s + 16 * p0.n_owned_s
            if npat_:
                moved_, kn_ = p0.n_owned_s + vecs_, ("csr_rowmask_kernel<double, RESID>" if masks_["entries"] else "csr_rowpat_kernel<double, RESID>")
            elif nval_:
                moved_, kn_ = 3 * op0.nnz + 4 * (p0.n_owned_s + 1) + vecs_, "csr_rowgather_kernel<double, RESID>"
            else:
                moved_, kn_ = 10 * op0.nnz + 4 * (p0.n_owned_s + 1) + vecs_, "csr_stream_kernel<double, RESID>"
            roofline = {"kernel": kn_ + " (rank 0's row shard of the fine level, r = b - A x, every range in one launch)", "bound": "hbm",
                        "achieved": round(moved_ / ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(moved_ / ms / 1e6 / HBM_PEAK_GBPS, 4),
                        "basis": "bytes the running operator format must stream (no counter pass at N > 1)",
                        "traffic": None, "bytes_per_launch": int(moved_), "ms_per_launch": round(ms, 5),
                        "bytes_csr_formula": int(by), "frac_csr_formula": round(by / ms / 1e6 / HBM_PEAK_GBPS, 4)}
            for d_ in (xs_, bs_, rs_):
                d_.free()
        # the same cycles by the resident single-GPU engine on rank 0 (the engine whose parity with the reference the
        # N = 1 line establishes): sharding must not change a bit of the iterates, the norms agree to all-reduce rounding
        check = None
        if rank == 0 and os.environ.get("PAMG_SHARD_CHECK", "1") != "0":
            try:
                d1 = DeviceMultilevelSolver(spec, device=local_rank, graph=not args.no_graph)
                xd, bd = capi.DeviceArray.from_host(x0), capi.DeviceArray.from_host(b)
                d1.load_device(xd, bd)
                r1 = np.asarray(d1.iterate_device(kpar))
                rel = np.abs(np.asarray(res_sh) - r1) / r1
                check = {"against": "the resident single-GPU engine on rank 0, same hierarchy / rhs", "cycles_compared": kpar,
                         "max_rel_diff": float(rel.max()), "tolerance": "1e-12 relative (only the all-reduced norm may differ)",
                         "ok": bool(rel.max() <= 1e-12)}
                d1.free(); xd.free(); bd.free()
            except Exception as e:                                  # noqa: BLE001
                check = {"error": repr(e)[:300]}
        if rank == 0:
            box["out"] = {
                "metric": "vcycle_iterations_per_sec", "value": round(args.steps / wall, 3), "unit": "cycles/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": wls["label"], "key": args.shard_workload, "n": n, "nnz": nnz, "levels": nlev, "cycle": "V(1,1)",
                           "parallelism": f"levels with >= {args.min_rows} rows row-sharded over {world} GPUs ({d2.sh.ns} levels), halo exchange "
                                          f"({info.get('transport', 'python schedule')}) overlapped with the interior rows, coarser levels replicated; "
                                          "hierarchy built once on rank 0 and shipped as arrays",
                           "hierarchy_setup": SETUP_NOTE,
                           "n1_point": "the N = 1 point of this strong-scaling curve is extra.c4 of the N = 1 line (the N = 1 headline is configs[2], "
                                       "whose order-exact Gauss-Seidel does not shard)"},
                "roofline": roofline, "cpu_baseline": None,
                "parity": {"sharded_vs_resident": check, "note": "parity of the resident engine with the reference on this workload: extra.c4 of the N = 1 line"},
                "sharded_driver": info,
                "host": {"setup_s": round(t_setup, 1), "partition_and_ship_s": round(t_ship, 1), "cores": os.cpu_count(), "setup": SETUP_NOTE},
                "residuals_gpu": [float(v) for v in res_sh],
            }
        del d2
        wd0.cancel()
        sys.excepthook = _excepthook
        # extras: N independent replicas of the Gauss-Seidel workload (what does not shard)
        wd = start_watchdog(float(os.environ.get("PAMG_EXTRAS_TIMEOUT", "600")), "the replica leg")
        if not args.no_extras:
            try:
                wlr = WORKLOADS[args.workload]
                A, ml, ts = build(wlr)
                br, xr = rhs(A.shape[0])
                dr = DeviceMultilevelSolver(ml, device=local_rank, graph=not args.no_graph)
                wr, _, resr, _, _ = time_resident(dr, br, xr, args.steps, args.warmup)
                dr.free()
                if rank == 0:
                    box["out"].setdefault("extra", {})["replicas"] = {
                        "workload": wlr["label"], "value": round(world * args.steps / wr, 3), "unit": "cycles/s", "scaling": "weak",
                        "ms_per_step": round(wr * 1e3 / args.steps, 4), "n_gpus": world,
                        "note": f"{world} independent replicas, one per GPU (order-exact Gauss-Seidel does not shard)"}
            except Exception as e:                                  # noqa: BLE001
                log(f"replica leg failed on rank {rank}: {e!r}")
                if rank == 0:
                    box["out"].setdefault("extra", {})["replicas"] = {"error": repr(e)[:300]}
        emit()
        dist.barrier()
        dist.destroy_process_group()
        wd.cancel()
        return box["out"]

    # =========================================================== main workload
    wl = WORKLOADS[args.workload]
    A, ml, t_setup = build(wl)
    t_assembly = assembly_s[0]
    n = A.shape[0]
    b, x0 = rhs(n)
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
    wall, ev_ms, res_gpu, xd, bd = time_resident(dml, b, x0, args.steps, args.warmup)

    # ---- roofline of the dominant bandwidth kernel: fine-level CSR SpMV (r = b - A x), HIP events
    #      on the solver's stream, algorithmic bytes of SURVEY.md 8(d)
    stream = dml.stream()
    A0 = dml.A[0]
    dml.store_device(xd)                              # the current iterate: realistic (non-zero) operand data
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
    bytes_resid = spmv_bytes(A.tocsr() if A.format != "csr" else A) + 8 * n               # + b read (scalar-CSR view)
    achieved = bytes_resid / spmv_ms / 1e6            # GB/s
    # operators with <= 256 distinct values (this stencil: 2) stream 8-bit value codes: the same launch with the values
    # themselves streamed (tune key 21 off) is timed beside it, so both numbers are in the line
    nvals = A0.value_codes()
    plain = None
    if nvals and rank == 0 and world == 1 and A.format == "csr":
        # a second resident copy of the fine-level operator on the plan of an operator WITHOUT value codes (LDS window 1536,
        # the staged kernel of rounds 1-2): what the same residual costs with 10 bytes per entry
        from pyamg_amd.hierarchy import sparse_op
        from pyamg_amd.multilevel import DeviceMatrix
        Ap_ = DeviceMatrix(sparse_op(A))
        Ap_.tune(val8=0, rowgather=0, lds_entries=1536, max_rows=1024)
        for _ in range(5):
            Ap_.spmv(capi.SPMV_RESID, xd, rd, b=bd, stream=stream)
        f0.record(stream)
        for _ in range(reps):
            Ap_.spmv(capi.SPMV_RESID, xd, rd, b=bd, stream=stream)
        f1.record(stream)
        f1.synchronize()
        ms_plain = f0.elapsed_ms(f1) / reps
        Ap_.free()
        plain = {"ms_per_launch": round(ms_plain, 5), "achieved": round(bytes_resid / ms_plain / 1e6, 1),
                 "frac": round(bytes_resid / ms_plain / 1e6 / HBM_PEAK_GBPS, 4),
                 "kernel": "csr_stream_kernel<double, RESID>: 16-bit column codes + the values as stored, LDS-staged products"}
    pmc = None
    if rank == 0 and world == 1 and not args.no_pmc and "grid" in wl and not wl.get("elasticity") and not wl.get("convdiff"):
        pmc = measure_traffic(wl["grid"], n)
    npats = A0.row_patterns()
    # measured ceiling of this device beside the datasheet peak
    ceiling = None
    if rank == 0:
        try:
            shapes = {k: round(capi.bandwidth_probe(k), 1) for k in ("copy1nt", "copy1", "copy8b", "copy", "triad", "read", "write")}
            ceiling = {"copy_GBps": max(shapes["copy1nt"], shapes["copy1"]), "triad_GBps": shapes["triad"], "shapes_GBps": shapes,
                       "how": "c = a over 1 GiB vectors, 20 launches (pamg_bandwidth_probe): copy1 / copy1nt = one 16-byte access per lane, no loop "
                              "(nt = nontemporal) -- the ceiling; copy8b = one 8-byte access per lane; copy / triad = grid-stride loops (what round 3 "
                              "quoted); read / write = one direction only"}
        except Exception as e:                                  # noqa: BLE001
            log(f"bandwidth probe failed: {e!r}")
    # bytes the format that actually ran has to move per launch (its operator stream + the vectors once)
    nnz0 = int(A.nnz)
    masks = A0.row_masks() if npats else {"entries": 0}
    if nvals and npats and masks["entries"]:
        streamed = 25 * n                             # one mask byte per row + b, x, r
        stream_note = (f"row masks: every row of this stencil is the interior row's list of {masks['entries']} (column - row, value) pairs with entries left "
                       f"out; a row streams ONE byte (which entries it has) instead of 12 bytes per stored entry + 4; offsets and values are launch constants"
                       + (f"; lattice form: 64 x 4 x {masks['planes_per_lane']} rows per workgroup, lines of {masks['line']}, planes of {masks['plane']} rows" if masks["lattice"] else ""))
    elif nvals and npats:
        streamed = 25 * n                             # one pattern byte per row + b, x, r; no entries, no row pointer (irregular rows: a few per mille)
        stream_note = (f"row patterns: {npats} lists of (column - row, value) pairs cover the rows of this stencil, a row streams ONE byte "
                       f"(its list number) instead of 12 bytes per stored entry + 4 ({nvals} distinct values)")
    elif nvals:
        streamed = bytes_resid - 9 * nnz0             # 2-byte column codes + 1-byte value codes instead of 4 + 8 bytes per entry
        stream_note = (f"16-bit column codes + 8-bit value codes ({nvals} distinct values): 3 instead of 12 bytes per stored entry reach the "
                       "kernel (lane = row: a gather instruction reads 64 consecutive values of x)")
    else:
        streamed = bytes_resid - 2 * nnz0             # 16-bit column codes: 10 instead of 12 bytes per stored entry
        stream_note = "16-bit window codes for the columns, values as stored: 10 bytes per stored entry"
    # the roofline fraction is taken on what MOVES: the larger of the bytes the running format must stream and the HBM
    # traffic counted in this run (a compressed format does not move the CSR formula's bytes; that figure stays beside it)
    moved = max(int(streamed), int(pmc["bytes_per_launch"]) if pmc else 0)
    kname = ((f"csr_rowmask3d_kernel<double, RESID, {masks['planes_per_lane']}>" if masks["lattice"] else "csr_rowmask_kernel<double, RESID>") if masks["entries"] else
             "csr_rowpat_kernel<double, RESID>" if npats else "csr_rowgather_kernel<double, RESID>" if nvals else "csr_stream_kernel<double, RESID>")
    structured = {"kernel": kname + " (fine-level r = b - A x)", "bound": "hbm",
                  "achieved": round(moved / spmv_ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                  "frac": round(moved / spmv_ms / 1e6 / HBM_PEAK_GBPS, 4),
                  "basis": "max(bytes the running operator format must stream, HBM traffic counted by rocprofv3 PMC in this run)",
                  "traffic": pmc["bytes_per_launch"] if pmc else None,
                  "bytes_per_launch": int(moved), "ms_per_launch": round(spmv_ms, 5),
                  "bytes_streamed_per_launch": int(streamed), "operator_stream": stream_note,
                  "bytes_csr_formula": int(bytes_resid), "achieved_csr_formula": round(achieved, 1),
                  "frac_csr_formula": round(achieved / HBM_PEAK_GBPS, 4),
                  "csr_formula_note": "SURVEY 8(d): 12 nnz + 4 (n + 1) + 8 n (x) + 8 n (r) + 8 n (b); exceeds the peak when the operator is streamed compressed"}
    if pmc:
        structured["traffic_detail"] = pmc
        structured["traffic_over_streamed"] = round(pmc["bytes_per_launch"] / max(streamed, 1), 3)
    if plain:
        # THE LEAD of the block: the north star's "fine-level CSR SpMV" -- the general kernel on the CSR arrays
        # (16-bit column codes + the values as stored, LDS-staged products), SURVEY 8(d)'s algorithmic bytes / its HIP-event time:
        # a fraction of the peak by construction.  The bit-identical structured form that runs inside the cycle on this stencil
        # (one mask byte per row) stands beside it under structured_*; its fraction is on the bytes that form must move.
        pmc_g = None
        if pmc is not None:                                   # the same two counter passes on the general kernel
            pmc_g = measure_traffic(wl["grid"], n, general=True)
        roofline = {"kernel": "csr_stream_kernel<double, RESID, npl2> (fine-level r = b - A x on the CSR arrays; 16-bit column codes + values as stored)",
                    "bound": "hbm", "achieved": plain["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": plain["frac"],
                    "traffic": pmc_g["bytes_per_launch"] if pmc_g else None,
                    "bytes_per_launch": int(bytes_resid), "ms_per_launch": plain["ms_per_launch"],
                    "bytes_streamed_per_launch": int(bytes_resid - 2 * nnz0),
                    "frac_on_streamed_bytes": round((bytes_resid - 2 * nnz0) / plain["ms_per_launch"] / 1e6 / HBM_PEAK_GBPS, 4),
                    "basis": "SURVEY 8(d): 12 nnz + 4 (n + 1) + 8 n (x) + 8 n (r) + 8 n (b), HIP events on the solver's stream",
                    "general_csr_ms": plain["ms_per_launch"], "general_csr_GBps": plain["achieved"], "general_csr_frac": plain["frac"],
                    "structured_kernel": structured["kernel"], "structured_ms_per_launch": structured["ms_per_launch"],
                    "structured_bytes_per_launch": structured["bytes_per_launch"], "structured_GBps": structured["achieved"],
                    "structured_frac": structured["frac"], "structured_traffic": structured["traffic"],
                    "structured_frac_csr_formula": structured["frac_csr_formula"], "structured": structured}
        if pmc_g:
            roofline["traffic_detail"] = pmc_g
            roofline["fetch_bytes"], roofline["write_bytes"] = pmc_g["fetch_corrected_x2"], pmc_g["write_size"]
            roofline["traffic_over_algorithmic"] = round(pmc_g["bytes_per_launch"] / bytes_resid, 3)
    else:
        roofline = structured
        if pmc:
            roofline["fetch_bytes"], roofline["write_bytes"] = pmc["fetch_corrected_x2"], pmc["write_size"]
    if ceiling:
        roofline["ceiling_copy_GBps"] = ceiling["copy_GBps"]
        roofline["ceiling"] = ceiling
        roofline["frac_of_ceiling"] = round(roofline["achieved"] / max(ceiling["copy_GBps"], 1.0), 4)

    # ---- the order-exact sweeps: latency-bound by the dependency chain of the reference's row order
    #      (levels of the schedule), not by HBM -- reported beside the bandwidth roofline so that the
    #      cycle time can be read: one forward Gauss-Seidel sweep per level, HIP events
    def sweep_table(dml_):
        rows_ = []
        for i, (L, dA) in enumerate(zip(ml.levels[:-1], dml_.A)):
            ni = L.A.shape[0]
            inf = dA.info()
            if not inf["gs_levels_fwd"]:
                continue
            xs_ = capi.DeviceArray.from_host(np.random.RandomState(i).rand(ni))
            bs_ = capi.DeviceArray.from_host(np.random.RandomState(100 + i).rand(ni))
            for _ in range(2):
                dA.gauss_seidel(xs_, bs_, sweep="forward", stream=stream)
            g0, g1 = capi.Event(), capi.Event()
            g0.record(stream)
            for _ in range(5):
                dA.gauss_seidel(xs_, bs_, sweep="forward", stream=stream)
            g1.record(stream)
            g1.synchronize()
            ms = g0.elapsed_ms(g1) / 5
            Ai = L.A.tocsr() if L.A.format != "csr" else L.A
            by = 12 * int(Ai.nnz) + 4 * (ni + 1) + 24 * ni            # Jacobi / GS / SOR sweep
            lane, tile, line, lanem = dA.lane_info(0), dA.tile_info(0), dA.line_info(0), dA.lanem_info(0)
            sched = (f"merged lane-parallel fast order: up to {lanem['s_max']} dependency levels eliminated into one super-level, {lanem['super_levels']} hand-offs, "
                     f"{lanem['units'] / max(1, lanem['rows']):.2f} x 64 operand slots per row, {lanem['launch_grid']} workgroups" if lanem["rows"] and dml_.order == "fast" and lanem["launch_grid"]
                     else f"line-scan fast order: {line['lines']} lines in {line['line_levels']} line levels, {line['launch_grid']} workgroups" if line["lines"] and dml_.order == "fast" and line["launch_grid"]
                     else f"lane-parallel fast order: {lane['lanes_per_row']} lanes per row x {lane['slots_per_lane']}, {lane['launch_grid']} workgroups" if lane["groups"] and dml_.order == "fast" and lane["launch_grid"]
                     else f"tiled exact sweep: {tile['tiles']} tiles" if tile["tiles"] else "granular / single-workgroup exact sweep")
            rows_.append({"level": i, "rows": int(ni), "nnz": int(Ai.nnz), "dependency_levels": int(inf["gs_levels_fwd"]), "scheduler": sched,
                          "hand_offs": int(lanem["super_levels"] if lanem["rows"] and dml_.order == "fast" and lanem["launch_grid"] else line["line_levels"] if line["lines"] and dml_.order == "fast" and line["launch_grid"] else inf["gs_levels_fwd"]),
                          "ms_per_forward_sweep": round(ms, 4), "us_per_dependency_level": round(1e3 * ms / inf["gs_levels_fwd"], 3),
                          "GBps": round(by / ms / 1e6, 1), "pct_of_hbm_peak": round(100 * by / ms / 1e6 / HBM_PEAK_GBPS, 2)})
            xs_.free(); bs_.free()
        return rows_

    sweeps = None
    if wl["smoother"] == GS and rank == 0:
        sweeps = sweep_table(dml)

    # ---- the same workload with ORDER-EXACT row sums (bit-identical to the reference; the fast order above keeps the sweep
    #      order and agrees to rounding): both modes are in the line
    exact_leg = None
    if wl["smoother"] == GS and rank == 0 and world == 1 and os.environ.get("PAMG_BENCH_EXACT", "1") != "0":
        try:
            dml_x = DeviceMultilevelSolver(ml, device=local_rank, graph=not args.no_graph, order="exact")
            wx, _, res_x, xdx, bdx = time_resident(dml_x, b, x0, args.steps, args.warmup)
            exact_leg = {"order": "exact", "value": round(args.steps / wx, 3), "unit": "cycles/s", "ms_per_step": round(wx * 1e3 / args.steps, 4),
                         "residuals_gpu": [float(v) for v in res_x], "gs_sweeps": sweep_table(dml_x),
                         "max_rel_diff_fast_vs_exact_norms": float(np.max(np.abs(np.asarray(res_x) - np.asarray(res_gpu)) / np.asarray(res_x)))}
            xdx.free(); bdx.free()
            dml_x.free()
            del dml_x
        except Exception as e:                                  # noqa: BLE001
            log(f"exact-order leg failed: {e!r}")
            exact_leg = {"error": repr(e)[:300]}

    if args.kernel_map and rank == 0:
        try:
            from tools.kernel_map import kernel_map
            sm = wl["smoother"]
            kind = sm if isinstance(sm, str) else sm[0]
            Path(args.kernel_map).write_text(json.dumps({"workload": args.workload, "peak_GBps": HBM_PEAK_GBPS, "ceiling": ceiling,
                                                         "entries": kernel_map(dml, kind)}, indent=1))
        except Exception as e:                                  # noqa: BLE001
            log(f"kernel map not written: {e!r}")

    ttt = None
    if rank == 0 and world == 1:
        try:
            ttt = time_to_tol(dml, b, x0)
        except Exception as e:                                  # noqa: BLE001
            ttt = {"error": repr(e)[:200]}

    accel_main = None
    if rank == 0 and world == 1 and args.cpu_cycles != 0 and not args.no_extras:
        try:
            accel_main = accel_leg(dml, ml, b, x0, "cg" if wl["smoother"] in (GS, JAC, CHEB) else "gmres", ref_iters=3 if n > 5_000_000 else 6)
        except Exception as e:                                  # noqa: BLE001
            accel_main = {"error": repr(e)[:200]}

    out = None
    cpu = parity = None
    if rank == 0 and world == 1 and args.cpu_cycles != 0:
        kcpu = args.cpu_cycles if args.cpu_cycles > 0 else (6 if n > 5_000_000 else 10)
        cpu, res_cpu = cpu_reference(ml, A, b, x0, kcpu)
        parity = parity_of(res_gpu, res_cpu)
        if args.protocol_cycles > 0:
            parity["reference_protocol"] = protocol_parity(dml, ml, n, k=args.protocol_cycles)
    setup_cmp = None
    if rank == 0 and world == 1 and not args.host_setup and not args.no_setup_compare and not wl.get("convdiff"):
        # the same setup by the reference alone: what the device setup operators buy, and what they change
        import scipy.sparse as sp
        A_r, ml_r, t_ref_setup = build(wl, device=False)
        sizes_d = [[int(L.A.shape[0]), int(L.A.nnz)] for L in ml.levels]
        sizes_r = [[int(L.A.shape[0]), int(L.A.nnz)] for L in ml_r.levels]
        setup_cmp = {"reference_setup_s": round(t_ref_setup, 1), "speedup": round(t_ref_setup / max(t_setup, 1e-9), 2),
                     "level_sizes_match": sizes_d == sizes_r}
        if sizes_d == sizes_r and len(ml.levels) > 1:
            d1 = abs(sp.csr_array(ml.levels[1].A) - sp.csr_array(ml_r.levels[1].A))
            setup_cmp["level1_A_max_rel_diff"] = float(d1.max() / abs(ml_r.levels[1].A).max())
        log(f"setup: reference alone {t_ref_setup:.1f}s, with the device setup operators {t_setup:.1f}s; {setup_cmp}")
        del A_r, ml_r
    if rank == 0:
        out = {
            "metric": "vcycle_iterations_per_sec", "value": round(world * args.steps / wall, 3), "unit": "cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["label"], "key": args.workload, "n": int(n), "nnz": int(A.nnz),
                       "levels": len(ml.levels), "cycle": "V(1,1)", "graph": not args.no_graph,
                       "parallelism": "1 GPU",
                       "hierarchy_setup": SETUP_NOTE + ("" if args.host_setup else
                                                        "; the CPU reference and the parity runs use this same hierarchy object"),},
            "event_ms_per_step": round(ev_ms / args.steps, 4),
            "spmv_GBps": roofline["achieved"], "spmv_pct_of_hbm_peak": round(100 * roofline["frac"], 2),
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            "host": {"setup_s": round(t_setup, 1), "upload_s": round(t_upload, 1), "operator_assembly_s": round(t_assembly, 1), "renumber_s": round(getattr(dml, "renumber_seconds", 0.0), 2),
                     "renumbered_levels": list(getattr(dml, "renumbered", [])), "cores": os.cpu_count(), "cpu_quota_cores": cpu_quota_cores(),
                     "setup": SETUP_NOTE, **(setup_cmp or {})},
            "residuals_gpu": [float(v) for v in res_gpu],
            "time_to_tol_1e-8": ttt,
        }
        out["config"]["gs_order"] = dml.order
        if accel_main:
            out["accel"] = accel_main
        if sweeps:
            # what bounds the STEP (not the bandwidth kernel above): the sweep kernel with the largest share, on SURVEY 8(d)'s bytes
            dom = max(sweeps, key=lambda r_: r_["ms_per_forward_sweep"])
            roofline["dominant_kernel"] = f"{dom['scheduler'].split(':')[0]}, level {dom['level']} ({dom['dependency_levels']} dependency levels)"
            roofline["dominant_kernel_ms"] = dom["ms_per_forward_sweep"]
            roofline["dominant_kernel_frac"] = round(dom["pct_of_hbm_peak"] / 100, 4)
            roofline["dominant_kernel_share_of_step"] = round(4 * dom["ms_per_forward_sweep"] / out["ms_per_step"], 3)
            roofline["dominant_kernel_us_per_dependency_level"] = dom["us_per_dependency_level"]
            roofline["sweeps_share_of_step"] = round(4 * sum(r_["ms_per_forward_sweep"] for r_ in sweeps) / out["ms_per_step"], 3)
        if parity:
            out["config"]["parity_random_b_max_abs_diff_over_r0"] = parity.get("max_abs_diff_over_r0")
            out["config"]["parity_random_b_ok"] = bool(parity.get("max_abs_diff_over_r0", 1.0) <= 1e-10)
            rp = parity.get("reference_protocol") or {}
            if rp:
                out["config"]["parity_protocol_ok"] = rp.get("ok")
                out["config"]["parity_protocol_max_rel_diff"] = rp.get("max_rel_diff")
                out["config"]["parity_protocol_cycles"] = rp.get("cycles_compared")
        if sweeps:
            out["gs_sweeps"] = {"note": "Gauss-Seidel in the reference's row order (same dependency DAG): one persistent launch per sweep, "
                                        "element-level hand-off; time = dependency levels x hand-off latency (a V(1,1) cycle with symmetric GS "
                                        "runs 4 sweeps per level).  order = fast: lane-parallel row sums and x 1/a_ii where the schedule fits "
                                        "(agrees with the reference to rounding); wide schedules keep the tiled exact sweep",
                                "order": dml.order, "per_level": sweeps}
        if exact_leg:
            out["exact_order"] = exact_leg
            if cpu and "residuals_gpu" in exact_leg:
                exact_leg["parity"] = parity_of(exact_leg["residuals_gpu"], res_cpu)
        if cpu:
            out["speedup_vs_cpu_reference"] = round(out["value"] / cpu["value"], 1)

    # The legs below are extras: whatever happens in them, the line of the main leg above must still be printed exactly
    # once.  Exceptions are recorded in the JSON; the watchdog prints the line and ends the process if the extras exceed
    # their time budget (a blocked HIP call cannot be interrupted from Python).
    box["out"] = out
    # a Chebyshev workload as the MAIN leg (--workload c4s | c4 | c4x): its modelled 2 / 4 / 8 GPU curve right here
    if rank == 0 and world == 1 and not args.no_model and args.workload in ("c4s", "c4", "c4x"):
        try:
            out["modelled_scaling"] = model_scaling(dml.spec, wl["label"], out["ms_per_step"])
            for r_ in out["modelled_scaling"]["rows"]:
                out["config"][f"modelled_ms_per_step_n{r_['n']}"] = r_["ms_per_step"]
        except Exception as e:                                        # noqa: BLE001
            log(f"scaling model failed: {e!r}")
            out["modelled_scaling"] = {"error": repr(e)[:300]}
    extras_budget = float(os.environ.get("PAMG_EXTRAS_TIMEOUT", "1500"))
    watchdog = start_watchdog(extras_budget, "the extra legs")

    # =========================================================== sharded leg (same hierarchy, Chebyshev)
    if not args.no_extras and args.workload in ("c3", "small"):
      try:
        dml.free()
        del dml
        np.random.seed(SEED)
        change_smoothers(ml, presmoother=CHEB, postsmoother=CHEB)     # rebinds smoothers, hierarchy unchanged
        spec = extract(ml)
        steps2 = max(args.steps, 10)
        d2 = DeviceMultilevelSolver(spec, device=local_rank, graph=not args.no_graph)
        w2, _, res2, _, _ = time_resident(d2, b, x0, steps2, args.warmup)
        d2.free()
        # the same cycles through the row-sharded C++ driver with ONE rank (no peers, so no exchange): what the driver's
        # own structure -- [owned | halo] vectors, split launches, its graph -- costs over the resident cycle
        drv = None
        try:
            from pyamg_amd.dist import DeviceOps, DistMultilevelSolver
            d2s = DistMultilevelSolver(spec, ops=DeviceOps(local_rank, spec.dtype), min_rows=args.min_rows)
            d2s.load(b, x0)
            res2s = d2s.iterate(6)
            d2s.load(b, x0)
            d2s.iterate(args.warmup, want_residuals=False)
            barrier()
            t0 = time.perf_counter()
            d2s.iterate(steps2)
            barrier()
            w2s = time.perf_counter() - t0
            drv = {"ms_per_step": round(w2s * 1e3 / steps2, 4), "value": round(steps2 / w2s, 3), "unit": "cycles/s",
                   "residual_norms_equal_resident": bool(np.array_equal(np.asarray(res2s), np.asarray(res2)[:len(res2s)])),
                   "max_rel_diff_vs_resident": float(np.max(np.abs(np.asarray(res2s) - np.asarray(res2)[:len(res2s)]) / np.asarray(res2)[:len(res2s)])),
                   **(d2s.native.info() if d2s.native is not None else {})}
            del d2s
        except Exception as e:                                  # noqa: BLE001
            drv = {"error": repr(e)[:300]}
        if rank == 0:
            glabel = "x".join(str(g) for g in wl["grid"])
            sh = {"workload": f"3D 7-pt Poisson {glabel} SA V-cycle, Chebyshev(3) smoother, fp64, 1 GPU (resident engine)",
                  "value": round(steps2 / w2, 3), "unit": "cycles/s", "ms_per_step": round(w2 * 1e3 / steps2, 4),
                  "steps": steps2, "scaling": "strong", "n_gpus": world,
                  "sharded_driver_one_rank": drv,
                  "residuals_gpu": [float(v) for v in res2]}
            if world == 1 and args.cpu_cycles != 0:
                c2cpu, r2cpu = cpu_reference(ml, A, b, x0, 3)
                sh["cpu_baseline"] = c2cpu
                sh["parity"] = parity_of(res2, r2cpu)
            out["sharded"] = sh
      except Exception as e:                                    # noqa: BLE001 -- extras never take the main line down
        log(f"sharded leg failed on rank {rank}: {e!r}")
        if rank == 0 and out is not None:
            out["sharded"] = {"error": repr(e)[:300]}

    # =========================================================== configs[1] leg (N = 1 only)
    if not args.no_extras and world == 1 and args.workload == "c3":
      try:
        wl2 = WORKLOADS["c2"]
        A2, ml2, ts2 = build(wl2)
        b2, x02 = rhs(A2.shape[0])
        d3 = DeviceMultilevelSolver(ml2, device=local_rank, graph=not args.no_graph)
        w3, _, res3, _, _ = time_resident(d3, b2, x02, 50, 5)
        ex = {"workload": wl2["label"], "value": round(50 / w3, 3), "unit": "cycles/s",
              "ms_per_step": round(w3 * 1e3 / 50, 4), "steps": 50, "host_setup_s": round(ts2, 1)}
        if args.cpu_cycles != 0:
            c3cpu, r3cpu = cpu_reference(ml2, A2, b2, x02, 10)
            ex["cpu_baseline"] = c3cpu
            ex["parity"] = parity_of(res3, r3cpu)
            ex["parity"]["reference_protocol"] = protocol_parity(d3, ml2, A2.shape[0])
        ex["time_to_tol_1e-8"] = time_to_tol(d3, b2, x02)
        if args.cpu_cycles != 0:
            try:
                ex["accel"] = accel_leg(d3, ml2, b2, x02, "cg", ref_iters=6)
                ex.update({k: v for k, v in ex["accel"].items() if k.startswith("hl_")})
            except Exception as e:                              # noqa: BLE001
                ex["accel"] = {"error": repr(e)[:200]}
        out["extra"] = {"c2": ex}
        d3.free()
      except Exception as e:                                    # noqa: BLE001
        log(f"configs[1] leg failed: {e!r}")
        out["extra"] = {"c2": {"error": repr(e)[:300]}}
      # configs[4] on one GPU: 3-D elasticity, BSR(3,3) -> (6,6); the SA default block Gauss-Seidel, then block Jacobi (the
      # smoother that shards) on the same hierarchy
      try:
        wl5 = WORKLOADS["c5"]
        A5, ml5, ts5 = build(wl5)
        b5, x05 = rhs(A5.shape[0])
        ex5 = {"workload": wl5["label"], "host_setup_s": round(ts5, 1), "levels": len(ml5.levels),
               "level_sizes": [[int(L.A.shape[0]), int(L.A.nnz), list(L.A.blocksize)] for L in ml5.levels][:3]}
        for tag, sm in (("block_gauss_seidel", None), ("block_jacobi", "block_jacobi")):
            if sm is not None:
                np.random.seed(SEED)
                change_smoothers(ml5, presmoother=sm, postsmoother=sm)
            d5 = DeviceMultilevelSolver(ml5, device=local_rank, graph=not args.no_graph)
            w5, _, res5, _, _ = time_resident(d5, b5, x05, 30, 3)
            leg = {"value": round(30 / w5, 3), "unit": "cycles/s", "ms_per_step": round(w5 * 1e3 / 30, 4), "steps": 30}
            if args.cpu_cycles != 0:
                c5cpu, r5cpu = cpu_reference(ml5, A5, b5, x05, 5)
                leg["cpu_baseline"] = c5cpu
                leg["parity"] = parity_of(res5, r5cpu)
                leg["parity"]["reference_protocol"] = protocol_parity(d5, ml5, A5.shape[0])
            leg["time_to_tol_1e-8"] = time_to_tol(d5, b5, x05)
            if tag == "block_jacobi":
                # roofline of the fine-level BSR(3,3) product r = b - A x (8 R C nblk + 4 nblk + 4 (n_brow + 1) + vectors)
                A5b = A5.tobsr(blocksize=(3, 3)) if A5.format != "bsr" else A5
                nblk5, nbrow5, n5 = int(A5b.indices.size), int(A5b.shape[0] // 3), int(A5b.shape[0])
                by5 = 8 * 9 * nblk5 + 4 * nblk5 + 4 * (nbrow5 + 1) + 8 * n5 + 8 * n5 + 8 * n5
                x5d, b5d, r5d = capi.DeviceArray.from_host(np.random.RandomState(1).rand(n5)), capi.DeviceArray.from_host(b5), capi.DeviceArray(n5, np.float64)
                st5 = d5.stream()
                for _ in range(5):
                    d5.A[0].spmv(capi.SPMV_RESID, x5d, r5d, b=b5d, stream=st5)
                h0, h1 = capi.Event(), capi.Event()
                h0.record(st5)
                for _ in range(50):
                    d5.A[0].spmv(capi.SPMV_RESID, x5d, r5d, b=b5d, stream=st5)
                h1.record(st5)
                h1.synchronize()
                ms5 = h0.elapsed_ms(h1) / 50
                ex5["roofline"] = {"kernel": "csr_stream_kernel<double, RESID> on the scalar view of the fine-level BSR(3,3) operator (entry order = SciPy's bsr_matvec)",
                                   "bound": "hbm", "bytes_per_launch": int(by5), "ms_per_launch": round(ms5, 5), "achieved": round(by5 / ms5 / 1e6, 1),
                                   "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(by5 / ms5 / 1e6 / HBM_PEAK_GBPS, 4),
                                   "note": f"{nblk5} blocks of 3x3 in {nbrow5} block rows: the operator is {8 * 9 * nblk5 / 1e6:.0f} MB, L2 / Infinity-Cache resident between launches"}
                for d_ in (x5d, b5d, r5d):
                    d_.free()
            d5.free()
            ex5[tag] = leg
        out.setdefault("extra", {})["c5"] = ex5
      except Exception as e:                                    # noqa: BLE001
        log(f"configs[4] leg failed: {e!r}")
        out.setdefault("extra", {})["c5"] = {"error": repr(e)[:300]}
      # the "next" row that used to lose to one CPU core: what pyamg.solve() configures for a non-symmetric operator --
      # energy-minimisation SA + gauss_seidel_nr (Kaczmarz) -- on 3-D upwind convection-diffusion 64^3; fast order of round 5 (csrc/pamg_kz.hip)
      try:
        wl6 = WORKLOADS["c6n3"]
        A6, ml6, ts6 = build(wl6)
        b6, x06 = rhs(A6.shape[0])
        d6 = DeviceMultilevelSolver(ml6, device=local_rank, graph=not args.no_graph)
        w6, _, res6, _, _ = time_resident(d6, b6, x06, 20, 3)
        ex6 = {"workload": wl6["label"], "value": round(20 / w6, 3), "unit": "cycles/s", "ms_per_step": round(w6 * 1e3 / 20, 4), "steps": 20,
               "host_setup_s": round(ts6, 1), "levels": len(ml6.levels), "gs_order": d6.order}
        if args.cpu_cycles != 0:
            c6cpu, r6cpu = cpu_reference(ml6, A6, b6, x06, 10)
            ex6["cpu_baseline"] = c6cpu
            ex6["speedup_vs_cpu_reference"] = round(ex6["value"] / c6cpu["value"], 1)
            ex6["parity"] = parity_of(res6, r6cpu)
            ex6["parity"]["reference_protocol"] = protocol_parity(d6, ml6, A6.shape[0])
        if args.cpu_cycles != 0:
            try:
                ex6["accel"] = accel_leg(d6, ml6, b6, x06, "gmres", ref_iters=6)      # what pyamg.solve() runs on a non-symmetric operator (blackbox.py:285-289)
                ex6.update({k: v for k, v in ex6["accel"].items() if k.startswith("hl_")})
            except Exception as e:                              # noqa: BLE001
                ex6["accel"] = {"error": repr(e)[:200]}
        out.setdefault("extra", {})["c6n3"] = ex6
        out["config"]["extra_c6n3_cycles_per_s"] = ex6["value"]
        out["config"]["extra_c6n3_vs_one_reference_core"] = ex6.get("speedup_vs_cpu_reference")
        d6.free()
      except Exception as e:                                    # noqa: BLE001
        log(f"gauss_seidel_nr leg failed: {e!r}")
        out.setdefault("extra", {})["c6n3"] = {"error": repr(e)[:300]}
      # f2's last smoother: overlapping Schwarz, 4 n dependency levels per sweep of an n x n grid -- one persistent launch per sweep since round 6
      try:
        wl8 = WORKLOADS["c8s"]
        A8, ml8, ts8 = build(wl8, device=False)          # the subdomains and inverted blocks are the reference's own (schwarz_parameters)
        b8, x08 = rhs(A8.shape[0])
        d8 = DeviceMultilevelSolver(ml8, device=local_rank, graph=not args.no_graph)
        w8, _, res8, _, _ = time_resident(d8, b8, x08, 20, 3)
        ex8 = {"workload": wl8["label"], "value": round(20 / w8, 3), "unit": "cycles/s", "ms_per_step": round(w8 * 1e3 / 20, 4), "steps": 20,
               "host_setup_s": round(ts8, 1), "levels": len(ml8.levels), "round5_ms_per_step_one_launch_per_level": 78.4}
        if args.cpu_cycles != 0:
            c8cpu, r8cpu = cpu_reference(ml8, A8, b8, x08, 6)
            ex8["cpu_baseline"] = c8cpu
            ex8["speedup_vs_cpu_reference"] = round(ex8["value"] / c8cpu["value"], 1)
            ex8["parity"] = parity_of(res8, r8cpu)
            ex8["parity"]["reference_protocol"] = protocol_parity(d8, ml8, A8.shape[0])
        out.setdefault("extra", {})["c8s"] = ex8
        d8.free()
      except Exception as e:                                    # noqa: BLE001
        log(f"Schwarz leg failed: {e!r}")
        out.setdefault("extra", {})["c8s"] = {"error": repr(e)[:300]}
      # configs[0]: the README's Ruge-Stuben example -- an irregular classical hierarchy at size, with the published
      # level sizes as the anchor (README.md:143-151)
      try:
        wl1 = WORKLOADS["c1"]
        A1, ml1, ts1 = build(wl1)
        b1, x01 = rhs(A1.shape[0])
        d1 = DeviceMultilevelSolver(ml1, device=local_rank, graph=not args.no_graph)
        w1, _, res1, _, _ = time_resident(d1, b1, x01, 50, 5)
        sizes = [[int(L.A.shape[0]), int(L.A.nnz)] for L in ml1.levels]
        ex1 = {"workload": wl1["label"], "value": round(50 / w1, 3), "unit": "cycles/s", "ms_per_step": round(w1 * 1e3 / 50, 4),
               "steps": 50, "host_setup_s": round(ts1, 1), "levels": len(ml1.levels), "level_sizes": sizes[:4],
               "readme_anchor": {"levels": 9, "level0": [250000, 1248000], "level1": [125000, 1121002],
                                 "match": bool(len(ml1.levels) == 9 and sizes[0] == [250000, 1248000] and sizes[1] == [125000, 1121002])}}
        if args.cpu_cycles != 0:
            c1cpu, r1cpu = cpu_reference(ml1, A1, b1, x01, 10)
            ex1["cpu_baseline"] = c1cpu
            ex1["parity"] = parity_of(res1, r1cpu)
            ex1["parity"]["reference_protocol"] = protocol_parity(d1, ml1, A1.shape[0])
        ex1["time_to_tol_1e-8"] = time_to_tol(d1, b1, x01)
        out.setdefault("extra", {})["c1"] = ex1
        d1.free()
      except Exception as e:                                    # noqa: BLE001
        log(f"configs[0] leg failed: {e!r}")
        out.setdefault("extra", {})["c1"] = {"error": repr(e)[:300]}
      # configs[3] at its own size on one GPU: 512^3 Chebyshev(3) (134 M rows, ~25 GB resident; the hierarchy is a ~1.5 min
      # build under device_setup, ~16 min by the reference alone); 384^3 only if that fails.  One reference run serves both
      # the parity protocol (b = 0, x0 = rand, 3 cycles, 1e-10 relative) and the CPU baseline (a 512^3 reference cycle is ~45 s).
      # Last leg: if it overruns, everything above is out.
      if args.workload == "c3":
        ex4 = None
        for key4 in ("c4x", "c4"):
            try:
                wl4 = WORKLOADS[key4]
                A4, ml4, ts4 = build(wl4)
                b4, x04 = rhs(A4.shape[0])
                t0 = time.time()
                d4 = DeviceMultilevelSolver(ml4, device=local_rank, graph=not args.no_graph)
                tu4 = time.time() - t0
                log(f"configs[3] leg ({key4}): n={A4.shape[0]} setup {ts4:.1f}s upload {tu4:.1f}s")
                w4, _, res4, _, _ = time_resident(d4, b4, x04, 10, 3)
                log(f"configs[3] leg ({key4}): {w4 * 100:.2f} ms per step")
                ex4 = {"workload": wl4["label"], "key": key4, "value": round(10 / w4, 3), "unit": "cycles/s", "ms_per_step": round(w4 * 1e3 / 10, 4),
                       "steps": 10, "host_setup_s": round(ts4, 1), "upload_s": round(tu4, 1), "levels": len(ml4.levels),
                       "n": int(A4.shape[0]), "nnz": int(A4.nnz), "hbm_bytes": int(d4.stats()["hbm_bytes"]),
                       "residuals_gpu": [float(v) for v in res4]}
                if args.cpu_cycles != 0:
                    pp = protocol_parity(d4, ml4, A4.shape[0], k=3)
                    ex4["cpu_baseline"] = cpu_from_protocol(pp, A4, b4)
                    ex4["parity"] = {"reference_protocol": pp}
                ex4["time_to_tol_1e-8"] = time_to_tol(d4, b4, x04)
                spec4 = d4.spec
                d4.free()
                if not args.no_model and key4 == "c4x":
                    try:
                        ms_ = model_scaling(spec4, wl4["label"], ex4["ms_per_step"])
                        out["modelled_scaling"] = ms_
                        for r_ in ms_["rows"]:
                            out["config"][f"modelled_c4x_ms_per_step_n{r_['n']}"] = r_["ms_per_step"]
                            out["config"][f"modelled_c4x_exchange_ms_n{r_['n']}"] = r_["exchange_ms_not_overlapped"]
                        out["config"]["modelled_c4x_ms_per_step_n1_measured"] = ex4["ms_per_step"]
                        out["config"]["modelled_note"] = "MODEL: rank N//2 timed on one GPU + halo bytes / 153 GB/s + 12 us per exchange; DESIGN 6"
                    except Exception as e:                            # noqa: BLE001
                        log(f"scaling model failed: {e!r}")
                        out["modelled_scaling"] = {"error": repr(e)[:300]}
                break
            except Exception as e:                                    # noqa: BLE001
                log(f"configs[3] leg ({key4}) failed: {e!r}")
                ex4 = {"error": repr(e)[:300], "key": key4}
        out.setdefault("extra", {})["c4"] = ex4

    emit()
    watchdog.cancel()
    return out


if __name__ == "__main__":
    main()
