#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel_trace / counter_collection CSVs) into small text + json summaries that can be
committed under profiles/:

  kernel_stats_summary.txt   kernel x grid -> calls, total, average duration, share of the run
  kernel_roofline.txt        the same launches joined with bench.py's --kernel-map (which level / operator / role a launch
                             is, its algorithmic bytes by SURVEY.md 8(d) and the bytes its operator format streams):
                             GB/s, % of the 8 TB/s HBM peak and % of the measured copy ceiling -- per kernel of the cycle

    python tools/summarize_prof.py <profile dir> <workload> <tag>
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

out, wl, tag = Path(sys.argv[1]), sys.argv[2], sys.argv[3]
BKIND = ["BLK_JACOBI", "BLK_GS", "PNT_JACOBI", "PNT_GS"]
EPI = ["SET", "ACC", "RESID", "AXPBY", "ACC_AXPBY", "SUMSQ", "ACCSEQ", "JACOBI", "JACOBI_B", "GS", "GS_B", "SOR", "JACOBI_IDX"]


def short(name):
    m = re.search(r"csr_stream_kernel<(\w+), *(\d+), *(\d+)(?:, *(\w+))?>", name)
    if m:
        return f"csr_stream<{m.group(1)},{EPI[int(m.group(2))]},npl{m.group(3)}>"
    m = re.search(r"csr_(rowgather|rowpat)_kernel<(\w+), *(\d+)>", name)
    if m:
        return f"csr_{m.group(1)}<{m.group(2)},{EPI[int(m.group(3))]}>"
    m = re.search(r"csr_rowmask3d_kernel<(\w+), *(\d+), *(\d+), *(\w+)(?:, *\d+)?>", name)
    if m:
        return f"csr_rowmask3d<{m.group(1)},{EPI[int(m.group(2))]},kz{m.group(3)}>"
    m = re.search(r"csr_rowmask_sumsq_kernel<(\w+), *(\d+)>", name)
    if m:
        return f"csr_rowmask<{m.group(1)},SUMSQ,nu{m.group(2)}>"
    m = re.search(r"csr_rowmask_kernel<(\w+), *(\d+), *(\d+), *(\w+)>", name)
    if m:
        return f"csr_rowmask<{m.group(1)},{EPI[int(m.group(2))]},nu{m.group(3)}>"
    m = re.search(r"gs_lane_kernel<(\w+), *(\d+), *(\d+), *(\d+), *(\w+)>", name)
    if m:
        return f"gs_lane<{m.group(1)},{EPI[int(m.group(2))]},L{m.group(3)},K{m.group(4)},{'oneXCD' if m.group(5) in ('true', '1') else 'chip'}>"
    m = re.search(r"gs_lanem\w*_kernel<(\w+)(?:, *(\d+))?(?:, *(\d+))?>", name)
    if m:
        return (f"gs_lanem<double,GS,{'oneXCD' if m.group(1) in ('true', '1') else 'chip'}" + (f",rpw{m.group(2)}" if m.group(2) else "")
                + (f",regs{m.group(3)}" if m.group(3) else "") + ">")
    m = re.search(r"gs_line_kernel<(\w+), *(\d+), *(\d+)>", name)
    if m:
        return f"gs_line<{m.group(1)},{EPI[int(m.group(2))]},K{m.group(3)}>"
    m = re.search(r"bsr_lane_kernel<(\d+), *(\d+), *(\d+), *(\d+)(?:, *\w+)?>", name)
    if m:
        return f"bsr_lane<double,BLK_GS,bs{m.group(1)},L{m.group(2)},K{m.group(3)},{'oneXCD' if m.group(4) == '1' else 'chip'}>"
    m = re.search(r"bsr_(gran|small|flow|stream)_kernel<(\w+), *(\d+)(?:, *(\w+))?>", name)
    if m:
        third = m.group(4)
        return f"bsr_{m.group(1)}<{m.group(2)},{BKIND[int(m.group(3))]}" + (f",{'bs' + third if third.isdigit() else third}" if third else "") + ">"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("pamg::", "")
    return re.sub(r"\(.*", "", name)[:70]


def family(k):
    if k.startswith("csr_"):
        m = re.search(r",(\w+?)(,npl\d|,kz\d|,nu\d)?>", k)
        return "csr", (m.group(1) if m else None)
    if k.startswith("gs_lanem"):
        return "gs_lanem", None
    if k.startswith("gs_lane"):
        return "gs_lane", None
    if k.startswith("gs_line"):
        return "gs_line", None
    if k.startswith("gs_tile"):
        return "gs_tile", None
    if k.startswith("gs_gran") or k.startswith("gs_flow"):
        return "gs_gran", None
    if k.startswith("bsr_lane"):
        return "bsr_lane", None
    if k.startswith("bsr_gran") or k.startswith("bsr_small") or k.startswith("bsr_flow"):
        m = re.search(r"<\w+,(\w+)(?:,bs(\d+))?", k)
        return "bsr_gs", (m.group(1), int(m.group(2)) if m and m.group(2) else None)
    if k.startswith("bsr_stream"):
        m = re.search(r"<\w+,(\w+)", k)
        return "bsr_stream", (m.group(1), None)
    return None, None


summary = {"workload": wl, "tag": tag}
kmap = None
for f in (out / "kernel_map.json",):
    if f.exists():
        kmap = json.loads(f.read_text())
# ---- kernel trace -> per (kernel, grid) count / total / avg
for f in glob.glob(str(out / "trace" / "**" / "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(lambda: [0, 0.0])
    durs = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k = short(r["Kernel_Name"])
            gs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 256)) or 256)
            durs[(k, gs // max(wg, 1))].append(d)
    # Two LEVELS of a hierarchy can run one kernel instantiation with one grid size (round 5: levels 2 and 3 both gs_lane<L32,K4,oneXCD> [256]; the
    # table then showed the mean of a 0.71 ms and a 0.10 ms sweep as "level 2").  Where the kernel map names several levels for a (kernel, grid),
    # its launches are split into that many groups at the largest gaps of their sorted durations (levels differ by factors in duration); the
    # longest group is the finest level.  Keys become (kernel, grid) or (kernel + ' #i', grid).
    if kmap:
        import math
        for (k, g) in list(durs):
            fam, epi = family(k)
            if fam not in ("gs_lane", "gs_lanem", "gs_line", "gs_tile", "gs_gran"):        # (block sweeps: the kernel name carries the block size of its level)
                continue
            lv = sorted({e["level"] for e in kmap["entries"] if e["family"] == fam and e.get("grid") == g})      # (entries that name THIS grid: a family without grids is told apart otherwise)
            if len(lv) < 2 or len(durs[(k, g)]) < 2 * len(lv):
                continue
            ds = sorted(durs.pop((k, g)), reverse=True)
            gaps = sorted(range(1, len(ds)), key=lambda i: math.log(ds[i - 1] / max(ds[i], 1e-9)), reverse=True)[:len(lv) - 1]
            cuts = [0] + sorted(gaps) + [len(ds)]
            for j in range(len(lv)):
                durs[(f"{k} #{j}", g)] = ds[cuts[j]:cuts[j + 1]]
    for key, v in durs.items():
        agg[key] = [len(v), sum(v)]
    tot = sum(v[1] for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = [f"{'kernel [workgroups]':62s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}"]
    for (k, g), (c, t) in rows:
        lines.append(f"{(k + ' [' + str(g) + ']'):62s} {c:8d} {t:12.1f} {t / c:10.2f} {100 * t / tot:6.2f}")
    (out / "kernel_stats_summary.txt").write_text("\n".join(lines) + "\n")
    summary["kernels"] = {f"{k} [{g}]": {"calls": c, "total_us": round(t, 1), "avg_us": round(t / c, 3)} for (k, g), (c, t) in rows}
    print("\n".join(lines[:14]))
    if kmap:
        peak = float(kmap.get("peak_GBps") or 8000.0)
        ceil = float((kmap.get("ceiling") or {}).get("copy_GBps") or 0.0)
        ents = kmap["entries"]
        rl = [f"per-kernel roofline of one {wl} cycle ({tag}); HBM peak {peak:.0f} GB/s" + (f", measured copy ceiling {ceil:.0f} GB/s" if ceil else ""),
              "bytes: SURVEY.md 8(d) algorithmic bytes of the launch | bytes the operator format that ran streams; GB/s on each",
              f"{'kernel [workgroups]':46s} {'lvl':>3s} {'op':>2s} {'role':44s} {'calls':>6s} {'avg_us':>9s} {'alg_MB':>9s} {'alg_GB/s':>9s} {'%peak':>6s} {'strm_MB':>9s} {'strm_GB/s':>9s} {'%peak':>6s} {'%ceil':>6s}"]
        table = []
        used = set()
        for (k, g), (c, t) in rows:
            fam, epi = family(k)
            if not fam:
                continue
            if fam in ("bsr_gs", "bsr_stream"):
                # block sweeps: the kernel name carries the flavour (and the compile-time block size); levels with one block size are told
                # apart by their grids -- the wider grid is the finer level (rows arrive sorted by total time, so order them here)
                kind, bs = epi
                same = sorted([(g2, k2) for (k2, g2), _ in rows if family(k2)[0] == fam and family(k2)[1] == epi and (fam == "bsr_gs" or True)], reverse=True)
                cand_all = [i for i, e in enumerate(ents) if e["family"] == fam and e.get("kind") == kind and (bs is None or e.get("bs") == bs)]
                if fam == "bsr_stream":
                    cand = [i for i in cand_all if ents[i]["grid"] == g] or cand_all
                else:
                    pos = same.index((g, k)) if (g, k) in same else 0
                    cand = cand_all[pos:pos + 1] or cand_all[-1:]
            else:
                cand = [i for i, e in enumerate(ents) if e["family"] == fam and (fam != "csr" or e["epi"] == epi) and (e["grid"] is None or e["grid"] == g or (fam == "csr" and 0 <= g - e["grid"] < 8))]
            if not cand:
                continue
            # several operators with one grid size (tiny levels): the first unused entry; launches split by duration ('kernel #j') take the
            # entries of the j-th level among the candidates
            mj = re.search(r" #(\d+)$", k)
            if mj:
                lvs = sorted({ents[i]["level"] for i in cand})
                want = lvs[min(int(mj.group(1)), len(lvs) - 1)]
                cand = [i for i in cand if ents[i]["level"] == want] or cand
            i = next((i for i in cand if i not in used), cand[0])
            used.add(i)
            e = ents[i]
            avg = t / c
            a_gbs = e["bytes_alg"] / avg / 1e3
            s_gbs = e["bytes_streamed"] / avg / 1e3 if e.get("bytes_streamed") else None
            rec = {"kernel": k, "grid": g, "level": e["level"], "op": e["op"], "role": e["what"], "calls": c, "avg_us": round(avg, 2), "bytes_alg": e["bytes_alg"],
                   "GBps_alg": round(a_gbs, 1), "pct_peak_alg": round(100 * a_gbs / peak, 2), "bytes_streamed": e.get("bytes_streamed"),
                   "GBps_streamed": round(s_gbs, 1) if s_gbs else None, "pct_peak_streamed": round(100 * s_gbs / peak, 2) if s_gbs else None,
                   "pct_ceiling_streamed": round(100 * s_gbs / ceil, 2) if (s_gbs and ceil) else None, "format": e.get("format")}
            if e.get("dependency_levels"):
                rec["us_per_dependency_level"] = round(avg / e["dependency_levels"], 3)
            table.append(rec)
            rl.append(f"{(k + ' [' + str(g) + ']')[:46]:46s} {e['level']:3d} {e['op']:>2s} {e['what'][:44]:44s} {c:6d} {avg:9.2f} {e['bytes_alg'] / 1e6:9.2f} {a_gbs:9.1f} {100 * a_gbs / peak:6.2f} "
                      + (f"{e['bytes_streamed'] / 1e6:9.2f} {s_gbs:9.1f} {100 * s_gbs / peak:6.2f} " + (f"{100 * s_gbs / ceil:6.2f}" if ceil else f"{'':6s}") if s_gbs else f"{'-':>9s} {'-':>9s} {'-':>6s} {'-':>6s}")
                      + (f"   {rec['us_per_dependency_level']} us per dependency level x {e['dependency_levels']}" if e.get("dependency_levels") else ""))
        # coverage: launches that belong to the cycle = kernels launched at least once per timed iteration (the level-0 convergence-check norm
        # runs once per iteration); share of their time that the table explains
        iters = max([r_["calls"] for r_ in table if "convergence-check" in r_["role"]] or [1])
        SETUP = ("bw_", "__amd_rocclr", "spg_", "lane_fill_kernel", "line_fill_kernel", "arn_", "tr_", "strength", "agg_", "fit_", "pinv", "csr_sub", "scale_rows", "reduce_")
        in_cycle = sum(t for (k, g), (c, t) in rows if c >= iters and not k.startswith(SETUP))
        mapped = sum(r_["calls"] * r_["avg_us"] for r_ in table if r_["calls"] >= iters)
        # the cycle's helper launches (hand-off buffer fills of the sweeps, vector updates of the polynomial smoother): no operator behind them
        helpers = [((k, g), (c, t)) for (k, g), (c, t) in rows if c >= iters and not k.startswith(SETUP) and family(k)[0] is None]
        if helpers:
            rl.append("helper launches of the cycle (no operator: sentinel fills of the sweeps' hand-off buffers, vector updates) -- time only:")
            for (k, g), (c, t) in helpers[:12]:
                rl.append(f"  {(k + ' [' + str(g) + ']')[:60]:60s} {c:6d} calls {t / c:9.2f} us")
        rl.append(f"coverage: the operator rows above explain {100 * mapped / max(in_cycle, 1e-9):.1f} % of the time of the kernels launched at least once per iteration ({iters} iterations in the trace); with the helper launches {100 * (mapped + sum(t for _, (c, t) in helpers)) / max(in_cycle, 1e-9):.1f} %")
        summary["kernel_roofline_coverage_pct"] = round(100 * mapped / max(in_cycle, 1e-9), 1)
        (out / "kernel_roofline.txt").write_text("\n".join(rl) + "\n")
        summary["kernel_roofline"] = table
        print("\n".join(rl[:24]))
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
