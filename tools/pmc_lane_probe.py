#!/usr/bin/env python3
"""What loads the memory system in the fast-order sweeps?  rocprofv3 --pmc passes (one small counter group each, kernel trace
only) over tools/lane_pmc.py; per kernel family (gs_lane by grid size, gs_line) the average counter values per launch.  Not product code."""
import csv, glob, json, os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
levels = sys.argv[1] if len(sys.argv) > 1 else "1+2"
tune = sys.argv[2] if len(sys.argv) > 2 else "{}"
tag = sys.argv[3] if len(sys.argv) > 3 else "base"
groups = [["GRBM_GUI_ACTIVE", "SQ_WAVES_sum", "SQ_BUSY_CYCLES"],
          ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum"],
          ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum", "TCP_TCC_NC_READ_REQ_sum", "TCP_TCC_UC_READ_REQ_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
          ["TCC_EA0_WRREQ_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCC_EA0_RDREQ_32B_sum"],
          ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"]]
out = {}
for g in groups:
    d = tempfile.mkdtemp(prefix="pmcl_")
    cmd = ["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, str(ROOT / "tools" / "lane_pmc.py"), levels, tune]
    try:
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
    except Exception as e:      # noqa: BLE001
        out[",".join(g)] = repr(e)
        continue
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                kn = r.get("Kernel_Name", "")
                if "gs_lane" in kn or "gs_line" in kn:
                    key = kn.split("(")[0][-70:] + " grid=" + str(r.get("Grid_Size", r.get("Grid_Size_X", "?")))
                    acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        for k, v in cs.items():
            out.setdefault(key, {})[k] = round(sum(v) / len(v), 1)
            out[key]["launches"] = len(v)
    print(json.dumps(out), flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"pmc_lane_probe_{tag}.json").write_text(json.dumps(out, indent=1))
