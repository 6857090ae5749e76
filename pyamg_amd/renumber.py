"""Blob-by-blob numbering of the interior levels of a hierarchy (host side; speed only, results bit-identical).

The reference numbers the unknowns of a coarse level in the order its aggregation met them (aggregation/aggregate.py
standard_aggregation -> amg_core/smoothed_aggregation.h:49-160): along the fine rows.  The neighbours of a coarse unknown
in the other grid directions are then a plane of aggregates away, a row range of A_l gathers x through ~1 000 distinct
columns per 1 536 entries and the gather is 40 % of the product.  The level-l unknowns are the solver's
own -- multilevel.py:566-662 never hands a level-l vector to the caller -- so the DEVICE copy of the hierarchy may number
them differently:

    A_l' = Pi A_l Pi^T     P_{l-1}' = P_{l-1} Pi^T     R_{l-1}' = Pi R_{l-1}     P_l' = Pi P_l     R_l' = R_l Pi^T

with rows moved and columns renamed only (csrc/pamg_renumber.hip): the entries of a row keep their stored order, so every
row sum adds the reference's products in the reference's order and the level-0 iterates are the reference's bit for bit.

The order is algebraic: level-l unknowns are grouped by the level-(l+1) aggregate they fall into (the column of the largest
entry of their row of P_l), those groups by their level-(l+2) aggregate, and so on -- the hierarchy's own aggregates are
compact blobs, so this is a space-filling order that needs no geometry (measured against a Morton order of the aggregates'
centroids: profiles/r06_microbench_renumber_orders.json).

Only levels whose smoothers do not depend on the numbering are touched: Jacobi and polynomial smoothers (row-wise formulas).
Gauss-Seidel / SOR / Kaczmarz / Schwarz / block sweeps visit the rows in the reference's order and Krylov smoothers add
inner products in it -- those levels keep the reference's numbering.
"""
from __future__ import annotations

import dataclasses
import os
from typing import List, Optional

import numpy as np

from . import _capi as capi
from .hierarchy import HierarchySpec, LevelSpec, SparseOp

ORDER_FREE = ("none", "jacobi", "polynomial")
MIN_ROWS = 32768           # below this the operator sits in L2 whatever its numbering


def _plain_csr(op: Optional[SparseOp]) -> bool:
    # scalar rows: CSR, or the BSR(1, 1) the reference's SA levels are stored in (same arrays, same arithmetic)
    return op is not None and op.fmt in ("csr", "bsr") and tuple(op.blocksize) == (1, 1)


def eligible(spec: HierarchySpec, l: int, min_rows: int = MIN_ROWS) -> bool:
    """level l is interior, plain CSR on both sides, large enough, and its smoothers are row-wise formulas"""
    nlev = len(spec.levels)
    if l < 1 or l > nlev - 2:
        return False
    L, up = spec.levels[l], spec.levels[l - 1]
    if not all(_plain_csr(o) for o in (L.A, L.P, L.R, up.P, up.R)):
        return False
    if L.A.shape[0] < min_rows or L.A.shape[0] != L.A.shape[1]:
        return False
    return all(s is None or s.kind in ORDER_FREE for s in (L.pre, L.post))


def _group_of(P: SparseOp) -> np.ndarray:
    """the next-level aggregate of every unknown: column of the largest |entry| of its row of P (empty rows: group 0)"""
    n = P.shape[0]
    out = np.empty(n, np.int32)
    ip, ix, dx = (np.ascontiguousarray(P.indptr, np.int32), np.ascontiguousarray(P.indices, np.int32), np.ascontiguousarray(P.data))
    capi.check(capi.load().pamg_csr_row_argmax_abs(capi.dtype_code(P.dtype), n, capi.ptr(ip), capi.ptr(ix), capi.ptr(dx), capi.ptr(out)),
               "pamg_csr_row_argmax_abs")
    np.maximum(out, 0, out=out)
    return out


def nested_orders(spec: HierarchySpec, levels: List[int]) -> dict:
    """{l: old_of_new (int32)} for the requested levels: one pass from the coarsest level up -- a level's unknowns are sorted by the
    rank of their next-level aggregate (that rank being the aggregate's place in ITS nested order), ties in the reference's order"""
    nlev = len(spec.levels)
    if not levels:
        return {}
    top = min(levels)
    rank = None                                     # place of every unknown of level k + 1 in its nested order (None = identity)
    orders = {}
    for k in range(nlev - 2, top - 1, -1):
        P = spec.levels[k].P
        if not _plain_csr(P):
            rank = None
            continue
        g = _group_of(P)
        key = g if rank is None else rank[g]
        old_of_new = np.argsort(key, kind="stable").astype(np.int32)
        rank = np.empty(old_of_new.size, np.int32)
        rank[old_of_new] = np.arange(old_of_new.size, dtype=np.int32)
        if k in levels:
            orders[k] = old_of_new
    return orders


def renumber_op(op: SparseOp, row_old_of_new: Optional[np.ndarray], col_new_of_old: Optional[np.ndarray]) -> SparseOp:
    """rows of op in the order row_old_of_new, columns renamed through col_new_of_old (None = unchanged); stored order inside a row kept"""
    if row_old_of_new is None and col_new_of_old is None:
        return op
    m, n = op.shape
    ip, ix, dx = (np.ascontiguousarray(op.indptr, np.int32), np.ascontiguousarray(op.indices, np.int32), np.ascontiguousarray(op.data))
    Bp, Bj, Bx = np.empty(m + 1, np.int32), np.empty(ix.size, np.int32), np.empty(dx.size, dx.dtype)
    rp = None if row_old_of_new is None else np.ascontiguousarray(row_old_of_new, np.int32)
    cp = None if col_new_of_old is None else np.ascontiguousarray(col_new_of_old, np.int32)
    if (rp is not None and rp.size != m) or (cp is not None and cp.size != n):
        raise ValueError("renumber_op: permutation length does not match the operator")
    capi.check(capi.load().pamg_csr_renumber(capi.dtype_code(op.dtype), m, n, capi.ptr(ip), capi.ptr(ix), capi.ptr(dx), capi.ptr(rp), capi.ptr(cp),
                                             capi.ptr(Bp), capi.ptr(Bj), capi.ptr(Bx)), "pamg_csr_renumber")
    return SparseOp(op.fmt, op.shape, (1, 1), Bp, Bj, Bx, src_format=op.src_format)


def renumber_levels(spec: HierarchySpec, min_rows: Optional[int] = None):
    """(device copy of spec with its eligible interior levels renumbered, {level: old_of_new}).  spec itself is not modified; when no
    level is eligible it is returned as is."""
    if min_rows is None:
        min_rows = int(os.environ.get("PAMG_RENUMBER_MIN_ROWS", MIN_ROWS))
    if np.dtype(spec.dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
        return spec, {}
    levels = [l for l in range(1, len(spec.levels) - 1) if eligible(spec, l, min_rows)]
    orders = nested_orders(spec, levels)
    if not orders:
        return spec, {}
    inverse = {}
    for l, o in orders.items():
        inv = np.empty(o.size, np.int32)
        inv[o] = np.arange(o.size, dtype=np.int32)
        inverse[l] = inv
    out = []
    for i, L in enumerate(spec.levels):
        A = renumber_op(L.A, orders.get(i), inverse.get(i))
        P = renumber_op(L.P, orders.get(i), inverse.get(i + 1)) if L.P is not None else None
        R = renumber_op(L.R, orders.get(i + 1), inverse.get(i)) if L.R is not None else None
        out.append(dataclasses.replace(L, A=A, P=P, R=R))
    return dataclasses.replace(spec, levels=out), orders
