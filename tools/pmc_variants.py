#!/usr/bin/env python3
"""HBM fetch / write bytes per launch of the fine-level residual kernel for a few variants of the operator stream
(tools/spmv_pmc.py under rocprofv3 --pmc, one counter per pass).  Not product code."""
import csv, glob, json, os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
grid = sys.argv[1:4] if len(sys.argv) > 3 else ["256", "256", "256"]
variants = [a for a in sys.argv[4:]] or ["--val8=0", "--val8=1", "--val8=1 --flags=3", "--val8=1 --cap=2048", "--val8=1 --cap=2048 --flags=3", "--val8=0 --flags=3"]
out = {}
for var in variants:
    rec = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pmcv_")
        cmd = ["rocprofv3", "--pmc", cname, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, str(ROOT / "tools" / "spmv_pmc.py")] + grid + var.split()
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", text=True)
        rec["line"] = [l for l in p.stdout.splitlines() if l.startswith("ok")][-1:] or None
        tot, cnt = 0.0, 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == cname and any(k in r.get("Kernel_Name", "") for k in ("csr_stream", "csr_rowgather", "csr_rowpat")) and int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) >= 256 * 1024:
                        tot += float(r["Counter_Value"]); cnt += 1
        rec[cname] = (tot / cnt * 1024) if cnt else None       # KiB -> bytes
    if rec.get("FETCH_SIZE"):
        rec["fetch_corrected_x2_GB"] = round(2 * rec["FETCH_SIZE"] / 1e9, 4)
    if rec.get("WRITE_SIZE"):
        rec["write_GB"] = round(rec["WRITE_SIZE"] / 1e9, 4)
    out[var] = rec
    print(var, rec, flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "pmc_variants_r03.json").write_text(json.dumps(out, indent=1))
