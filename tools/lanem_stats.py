#!/usr/bin/env python3
"""Statistics of the merged lane plan (pamg_lanem_plan.h) on dumped operators (npy triplets): hand-offs, operands, units, K classes per s.
    python tools/lanem_stats.py /tmp/hier/p128_L1 [s ...] [--run]   (--run: replay the sweep on the CPU and compare with the sequential sweep)"""
import ctypes, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
lib = ctypes.CDLL(str(ROOT / "tests" / "build" / "lanem_emul.so"))
lib.lanem_emul_sweep_f64.restype = ctypes.c_int
pfx = sys.argv[1]
ss = [int(a) for a in sys.argv[2:] if not a.startswith("--")] or [1, 2, 3, 4]
run = "--run" in sys.argv
back = "--backward" in sys.argv
LEN = 512
RPW = 1
for a_ in sys.argv:
    if a_.startswith("--len="):
        LEN = int(a_[6:])
    if a_.startswith("--rpw="):
        RPW = int(a_[6:])
Ap, Aj, Ax = (np.load(f"{pfx}_{k}.npy") for k in ("indptr", "indices", "data"))
n = Ap.size - 1
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
rng = np.random.RandomState(0)
x0, b = rng.rand(n), rng.rand(n)
ref = None
if run:
    from oracle import oracle as orc
    ref = x0.copy()
    orc.gauss_seidel(Ap, Aj, Ax, ref, b, *( (n - 1, -1, -1) if back else (0, n, 1)))
for s in ss:
    st = np.zeros(16, dtype=np.int64); gs = np.zeros(2)
    x = x0.copy()
    t = time.time()
    rc = lib.lanem_emul_sweep_f64(n, p(Ap), p(Aj), p(Ax), p(x), p(b), *((n - 1, -1, -1) if back else (0, n, 1)), s, ctypes.c_double(1e3), LEN, p(st), p(gs), 0, 0 if run else 1, RPW)
    dt = time.time() - t
    msg = f"s={s}: rc={rc} levels {st[1]} -> super {st[0]}; rows {st[2]}; operands early/old/b per row {st[4]/st[2]:.1f}/{st[5]/st[2]:.1f}/{st[6]/st[2]:.2f} (direct {st[7]/st[2]:.1f}); units/row {st[3]/st[2]:.3f} K1,2,3,4+ {[int(v) for v in st[12:16]]}; max_len {st[8]} closed len/growth {st[9]}/{st[10]} growth {gs[0]:.2f}; widest {st[11]}; {dt:.1f}s"
    if run and rc == 0:
        msg += f"; vs sequential {np.max(np.abs(x - ref)) / np.max(np.abs(ref)):.2e}"
    print(msg, flush=True)
