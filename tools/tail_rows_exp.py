#!/usr/bin/env python3
"""bench.py --workload W with the hierarchy's tail in one launch (PAMG_TAIL=1) for several row limits.  Not product code."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
out = []
for wl in sys.argv[1:] or ["c2", "c1"]:
    for rows in (0, 64, 128, 256, 512, 2048):
        env = dict(os.environ, PAMG_TAIL="1" if rows else "0", PAMG_TAIL_ROWS=str(max(rows, 1)))
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", wl, "--no-extras", "--cpu-cycles", "0", "--no-pmc", "--no-setup-compare",
                            "--steps", "200", "--warmup", "10", "--protocol-cycles", "0"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rec = {"workload": wl, "tail_rows": rows, "ms_per_step": d["ms_per_step"], "value": d["value"]}
        except Exception as e:  # noqa: BLE001
            rec = {"workload": wl, "tail_rows": rows, "error": (r.stderr or repr(e))[-300:]}
        print(rec, flush=True)
        out.append(rec)
(ROOT / "gpurun_out" / "tail_rows_exp.json").write_text(json.dumps(out, indent=1))
