#!/usr/bin/env python3
"""Where does the fine-level residual kernel spend its cycles?  A few rocprofv3 --pmc passes (one small counter group each,
kernel trace only) over tools/spmv_pmc.py.  Not product code."""
import csv, glob, json, os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
grid = os.environ.get("PMC_GRID", "256,256,256").split(",")
LEVEL1 = "--level1" in sys.argv
variants = [a.replace("+", " ") for a in sys.argv[1:] if a != "--level1"] or (["level1"] if LEVEL1 else ["--val8=1", "--val8=0"])
GROUPS_SHORT = os.environ.get("PMC_SHORT", "0") == "1"
groups = [["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVES_sum"],
          ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"],
          ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
          ["TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_BUSY_avr", "TCC_TAG_STALL_sum"],
          ["SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
          ["SQ_INST_LEVEL_VMEM", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU"],
          ["MemUnitStalled", "VALUBusy", "LDSBankConflict", "OccupancyPercent"]]
if GROUPS_SHORT:
    groups = [groups[0], groups[1], groups[3]]
out = {}
for var in variants:
    rec = {}
    for g in groups:
        d = tempfile.mkdtemp(prefix="pmcs_")
        if LEVEL1:
            cmd = ["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, str(ROOT / "tools" / "spmv_pmc_level1.py")]
        else:
            cmd = ["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, str(ROOT / "tools" / "spmv_pmc.py")] + grid + var.split() + ["--launches=4"]
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        except Exception as e:      # noqa: BLE001
            rec[",".join(g)] = repr(e)
            continue
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if any(k in r.get("Kernel_Name", "") for k in ("csr_stream", "csr_rowgather", "csr_rowpat", "csr_rowmask")) and int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) >= (4096 * 256 if LEVEL1 else 256 * 1024):
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            rec[k] = round(sum(v) / len(v), 1)
    out[var] = rec
    print(var, json.dumps(rec), flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / ("pmc_stall_probe_level1.json" if LEVEL1 else "pmc_stall_probe.json")).write_text(json.dumps(out, indent=1))
