#!/usr/bin/env python3
"""Does a symmetric renumbering of the first SA level speed up its products?  (second half.)

The SA-level operators of the 3-D Poisson hierarchy gather x through scattered columns: a 1 536-entry range of A1 holds about 1 000 distinct columns
in about 140 runs.  A renumbering Pi of the coarse unknowns (A1' = Pi A1 Pi^T, R0' = Pi R0, P0' = P0 Pi^T; entries of a row stay in their stored
order, so every row sum is formed exactly as before) changes which columns a range touches.  This tool times the three products in the staged kernel
under several orderings:

  identity   the aggregation's own numbering
  rcm        reverse Cuthill-McKee of A1's graph (scipy)
  geo        aggregates sorted by the Morton code of their centroid (needs the grid: an upper bound for what an algebraic ordering could find)
  geolex     aggregates sorted lexicographically by centroid (z, y, x)
  nested     algebraic: coarse unknowns grouped by the aggregate they fall into on the NEXT level, those groups by the level after, ... (the
             hierarchy's own aggregates are compact blobs, so this is a space-filling order that needs no geometry)

and prints the locality of each (distinct columns and runs per 1 536 entries).  Not product code."""
import argparse, json, sys
from pathlib import Path
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401
import pyamg
from pyamg_amd import _capi as capi
from pyamg_amd.aggregation import device_setup
from pyamg_amd.hierarchy import extract, SparseOp
from pyamg_amd.multilevel import DeviceMatrix

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs="+", default=[256, 256, 256])
ap.add_argument("--tag", default="renumber")
ap.add_argument("--orders", nargs="+", default=["identity", "rcm", "nested", "geo", "geolex"])
a = ap.parse_args()
grid = tuple(a.grid)
A = pyamg.gallery.poisson(grid, format="csr")
np.random.seed(1)
with device_setup(pyamg):
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
spec = extract(ml)
A1 = spec.levels[1].A.to_scipy().tocsr()
P0 = spec.levels[0].P.to_scipy().tocsr()
R0 = spec.levels[0].R.to_scipy().tocsr()
nc = A1.shape[0]


def relabel_cols(M, new_of_old):
    """columns renamed, entries of each row left in their stored order"""
    return SparseOp("csr", M.shape, (1, 1), M.indptr.astype(np.int32), new_of_old[M.indices].astype(np.int32), M.data.copy())


def permute_rows(M, old_of_new):
    """rows reordered (row i of the result = row old_of_new[i]), entries in stored order"""
    cnt = np.diff(M.indptr)[old_of_new]
    ptr = np.zeros(M.shape[0] + 1, np.int64); np.cumsum(cnt, out=ptr[1:])
    src = np.repeat(M.indptr[:-1][old_of_new].astype(np.int64) - ptr[:-1], cnt) + np.arange(ptr[-1])
    return sp.csr_array((M.data[src], M.indices[src], ptr.astype(np.int32)), shape=M.shape)


def locality(op, cap=1536, samples=400):
    rng = np.random.RandomState(0)
    nnz = op.indices.size
    d, r = [], []
    for s in rng.randint(0, max(1, nnz - cap), samples):
        c = np.unique(op.indices[s:s + cap])
        d.append(c.size); r.append(1 + int(np.count_nonzero(np.diff(c) > 1)))
    return round(float(np.mean(d)), 1), round(float(np.mean(r)), 1)


def centroids():
    # aggregate of a fine point = the column of its (single-block) row of the tentative prolongator pattern: P0's largest entry
    absP = abs(P0)
    agg = np.asarray(absP.argmax(axis=1)).ravel()
    nz, ny, nx = grid
    idx = np.arange(A.shape[0])
    zc, yc, xc = idx // (ny * nx), (idx // nx) % ny, idx % nx
    cnt = np.maximum(np.bincount(agg, minlength=nc), 1)
    return [np.bincount(agg, weights=w, minlength=nc) / cnt for w in (zc, yc, xc)]


def morton(cz, cy, cx):
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return (spread(cz) << 2) | (spread(cy) << 1) | spread(cx)


out = []
for order in a.orders:
    if order == "identity":
        old_of_new = np.arange(nc)
    elif order == "rcm":
        old_of_new = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(A1), symmetric_mode=True)).astype(np.int64)
    elif order in ("geo", "geolex"):
        cz, cy, cx = centroids()
        if order == "geo":
            old_of_new = np.argsort(morton(np.rint(cz), np.rint(cy), np.rint(cx)), kind="stable")
        else:
            old_of_new = np.lexsort((cx, cy, cz))
    elif order.startswith("nested"):
        # nested = every level of grouping, groups in the order of their ids; nestedK = K levels of grouping, groups in the order of their FIRST member
        # (the aggregation numbers its aggregates along the fine rows, which is what P0's gather likes: keep that order between the groups)
        depth = int(order[6:]) if len(order) > 6 else 0
        ident = np.arange(nc)
        keys = [ident]
        g = ident
        for lv in spec.levels[1:-1][:depth or None]:
            if lv.P is None:
                break
            Pl = abs(lv.P.to_scipy().tocsr())
            g = np.asarray(Pl.argmax(axis=1)).ravel()[g]       # the aggregate (column of the largest entry of the row) one level further down
            if depth:
                first = np.full(int(g.max()) + 1, nc); np.minimum.at(first, g, ident)
                keys.append(first[g])
            else:
                keys.append(g)
        old_of_new = np.lexsort(tuple(keys))
    else:
        raise SystemExit(order)
    new_of_old = np.empty(nc, np.int64); new_of_old[old_of_new] = np.arange(nc)
    ops = [("A1", relabel_cols(permute_rows(A1, old_of_new), new_of_old), capi.SPMV_RESID),
           ("P0", relabel_cols(P0, new_of_old), capi.SPMV_SET),
           ("R0", SparseOp("csr", R0.shape, (1, 1), *(lambda M: (M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data))(permute_rows(R0, old_of_new))), capi.SPMV_SET)]
    for name, op, epi in ops:
        m, n = op.shape
        dA = DeviceMatrix(op)
        dA.autotune()
        rng = np.random.RandomState(0)
        x = capi.DeviceArray.from_host(rng.rand(n)); b = capi.DeviceArray.from_host(rng.rand(m)); y = capi.DeviceArray(m, np.float64)
        kw = dict(b=b) if epi == capi.SPMV_RESID else {}
        for _ in range(3):
            dA.spmv(epi, x, y, **kw)
        capi.sync()
        got = y.download()
        ref = (b.download() - op.to_scipy() @ x.download()) if epi == capi.SPMV_RESID else op.to_scipy() @ x.download()
        e0, e1 = capi.Event(), capi.Event()
        e0.record()
        for _ in range(20):
            dA.spmv(epi, x, y, **kw)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_ms(e1) / 20
        by = 12 * op.nnz + 4 * (m + 1) + 8 * n + 8 * m + (8 * m if epi == capi.SPMV_RESID else 0)
        dist, runs = locality(op)
        rec = {"order": order, "op": name, "shape": [m, n], "nnz": int(op.nnz), "ms": round(ms, 5), "alg_GBps": round(by / ms / 1e6, 1),
               "distinct_columns_per_1536": dist, "runs_per_1536": runs, "max_abs_err": float(np.max(np.abs(got - ref)))}
        print(rec, flush=True)
        out.append(rec)
        dA.free()
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"microbench_{a.tag}.json").write_text(json.dumps(out, indent=1))
