#!/usr/bin/env python3
"""which access shape gets closest to the HBM on this device (pamg_bandwidth_probe kinds)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyamg_amd import _capi as capi
out = {}
for n in (1 << 24, 1 << 27):
    for k in ("copy", "triad", "copy1", "copy4", "copy4nt", "copy8", "read", "write", "memcpy", "copy8b", "copy8bnt", "copy1nt"):
        out[f"{k}_n{n}"] = round(max(capi.bandwidth_probe(k, n, 20) for _ in range(3)), 1)
        print(k, n, out[f"{k}_n{n}"], flush=True)
(ROOT / "gpurun_out" / "microbench_bw.json").write_text(json.dumps(out, indent=1))
