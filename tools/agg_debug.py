import sys, time, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport, pyamg
from pyamg import amg_core
from pyamg_amd import amg_core as gcore
from pyamg_amd.aggregation import device_setup
import scipy.sparse as sp
for g in [(16,16,16),(64,64,64),(128,128,128),(200,200,200)]:
    A = pyamg.gallery.poisson(g, format="csr")
    np.random.seed(0)
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, keep=True)
    for l, L in enumerate(ml.levels[:-1]):
        C = sp.csr_array(L.C)
        n = C.shape[0]
        x, y = np.empty(n, np.int32), np.empty(n, np.int32)
        t = time.time(); cr = amg_core.standard_aggregation(n, C.indptr.astype(np.int32), C.indices.astype(np.int32), x, y); tr = time.time() - t
        xd, yd = np.empty(n, np.int32), np.empty(n, np.int32)
        try:
            t = time.time(); cd = gcore.standard_aggregation(n, C.indptr.astype(np.int32), C.indices.astype(np.int32), xd, yd); td = time.time() - t
            print(g, l, n, C.nnz, "ref %.3fs dev %.3fs" % (tr, td), cr == cd, np.array_equal(x, xd), np.array_equal(y[:cr], yd[:cd]), flush=True)
        except Exception as e:
            print(g, l, n, C.nnz, "FAILED", repr(e)[:200], flush=True)
