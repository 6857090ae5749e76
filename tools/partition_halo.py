#!/usr/bin/env python3
"""Halo rows of an N-way partition of the levels of the SA hierarchy of 3-D Poisson: contiguous row blocks (what pyamg_amd.dist shards: z-slabs on the fine
grid, index ranges of the aggregates on level 1) against 2 x 2 x 2 BRICKS (rows assigned by grid coordinates; on level 1 by the centroid of the aggregate).
Counts, per part, the distinct columns of A_l outside the part (the halo A_l alone needs; P and R add to it in the real plan).  Host only (oracle/_ref):
a design aid for DESIGN 6 -- the sharded driver itself partitions by contiguous row blocks.    python tools/partition_halo.py 128 [N]"""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle.refimport  # noqa: F401
import pyamg

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
A = pyamg.gallery.poisson((n1, n1, n1), format="csr")
np.random.seed(1)
ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, keep=True)
g = np.arange(n1)
Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
coords = [np.stack([Z.ravel(), Y.ravel(), X.ravel()], axis=1).astype(np.float64)]
for l in range(len(ml.levels) - 1):
    Agg = ml.levels[l].AggOp.tocsr()                       # fine x coarse, one 1 per row
    cnt = np.asarray(Agg.sum(axis=0)).ravel()
    coords.append((Agg.T @ coords[l]) / np.maximum(cnt, 1)[:, None])


def halo(M, part):
    M = M.tocsr()
    rows = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
    pr, pc = part[rows], part[M.indices]
    out = []
    for p in range(part.max() + 1):
        sel = (pr == p) & (pc != p)
        out.append(int(np.unique(M.indices[sel]).size))
    return out


res = {"grid": n1, "parts": N, "levels": []}
for l in range(min(3, len(ml.levels))):
    M = ml.levels[l].A
    n = M.shape[0]
    if n < 20 * N:
        break
    slab = np.minimum((np.arange(n) * N) // n, N - 1)
    c = coords[l]
    k = round(N ** (1 / 3))
    if k ** 3 != N:
        break
    q = [np.minimum((np.argsort(np.argsort(c[:, d], kind="stable"), kind="stable") * k) // n, k - 1) for d in range(3)]   # equal thirds of the rows along every axis
    brick = (q[0] * k + q[1]) * k + q[2]
    hs, hb = halo(M, slab), halo(M, brick)
    owned_b = np.bincount(brick, minlength=N)
    res["levels"].append({"level": l, "rows": int(n), "owned_per_part_slabs": int(n // N), "halo_slabs_max": max(hs), "halo_slabs_mean": float(np.mean(hs)),
                          "halo_slabs_max_over_owned": round(max(hs) / (n / N), 4), "owned_bricks_min_max": [int(owned_b.min()), int(owned_b.max())],
                          "halo_bricks_max": max(hb), "halo_bricks_mean": float(np.mean(hb)), "halo_bricks_max_over_owned": round(max(hb) / (n / N), 4),
                          "neighbours_slabs": 2, "neighbours_bricks_up_to": 7 if l == 0 else 26})
    print(json.dumps(res["levels"][-1]), flush=True)
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / f"partition_halo_{n1}_N{N}.json").write_text(json.dumps(res, indent=1))
