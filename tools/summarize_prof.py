#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel_stats / kernel_trace / counter_collection CSVs) into
small text+json summaries that can be committed under profiles/."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

out, wl, tag = Path(sys.argv[1]), sys.argv[2], sys.argv[3]


def short(name):
    m = re.search(r"csr_stream_kernel<(\w+), *(\d+), *(\d+)>", name)
    epi = ["SET", "ACC", "RESID", "AXPBY", "ACC_AXPBY", "SUMSQ", "ACCSEQ", "JACOBI", "JACOBI_B", "GS", "GS_B", "SOR"]
    if m:
        return f"csr_stream<{m.group(1)},{epi[int(m.group(2))]},npl{m.group(3)}>"
    m = re.search(r"csr_(rowgather|rowpat)_kernel<(\w+), *(\d+)>", name)
    if m:
        return f"csr_{m.group(1)}<{m.group(2)},{epi[int(m.group(3))]}>"
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:70]


summary = {"workload": wl, "tag": tag}
# ---- kernel trace -> per-kernel count / total / avg
for f in glob.glob(str(out / "trace" / "**" / "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for r in csv.DictReader(fh):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k = short(r["Kernel_Name"])
            gs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            if gs >= 256 * 4096:                            # big launches (fine level): keep them apart
                k = f"{k} [grid {gs // 256} wg]"
            agg[k][0] += 1
            agg[k][1] += d
    tot = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = [f"{'kernel':55s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}"]
    for k, (c, t) in rows:
        lines.append(f"{k:55s} {c:8d} {t:12.1f} {t / c:10.2f} {100 * t / tot:6.2f}")
    (out / "kernel_stats_summary.txt").write_text("\n".join(lines) + "\n")
    summary["kernels"] = {k: {"calls": c, "total_us": round(t, 1), "avg_us": round(t / c, 3)} for k, (c, t) in rows}
    print("\n".join(lines[:14]))
# ---- PMC passes
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    sub = "pmc_fetch" if cname == "FETCH_SIZE" else "pmc_write"
    for f in glob.glob(str(out / sub / "**" / "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != cname:
                    continue
                k = short(r["Kernel_Name"])
                gs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
                if gs >= 256 * 4096:
                    k = f"{k} [grid {gs // 256} wg]"
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
        summary[cname] = {k: {"calls": c, "avg_value_KB": round(v / c, 1)} for k, (c, v) in
                          sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]}
(out / "summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps({k: summary.get(k) for k in ("FETCH_SIZE", "WRITE_SIZE")}, indent=1)[:3000])
