"""Row-sharded multigrid cycle: one process per GPU, ``torch.distributed`` (RCCL over xGMI).

What shards: SpMV / residual / restriction / prolongation, weighted Jacobi and
polynomial (Chebyshev, Richardson) smoothing -- independent rows plus ONE halo exchange of the
input vector per operator application.  Order-exact Gauss-Seidel does not shard (global
sequential dependency); hierarchies using it are rejected here (run replicas instead).

Layout.  Every level with at least ``min_rows`` unknowns is split into contiguous row blocks,
one per rank.  A rank holds its rows of ``A_l``, the rows of ``P_l`` it owns (fine rows) and
the rows of ``R_l`` it owns (coarse rows), all renumbered into a local column space
``[owned | halo]`` where the halo of level ``l`` is the union of the off-rank columns that
``A_l``, ``P_{l-1}`` and ``R_l`` touch, grouped by owning rank.  Every level-``l`` vector is a
buffer ``[owned | halo]``; one exchange fills the halo straight from the owners' packed send
buffers (point-to-point ``isend``/``irecv`` with the actual neighbours only -- xGMI is
point-to-point, a ring all-gather of x would move N times the data).  Per-row arithmetic is
unchanged (entries keep their storage order), so the sharded iterates are bit-identical to
the single-GPU ones; only norms (one all-reduced scalar per iteration) differ in the last bits.

Below the last sharded level the hierarchy is tiny (<2 % of the nonzeros): its right-hand
side is assembled on every rank by an all-reduce of disjoint slices and the remaining cycle
runs redundantly on every GPU with the resident single-GPU engine -- the "collapse to GPU 0"
schedule without the broadcast back.

Local arithmetic is delegated to an ``ops`` object: ``DeviceOps`` (HIP kernels through the C
ABI, this module) in production; the CPU tests inject an oracle-backed twin so the
partition / halo / cycle logic runs under ``gloo`` without a GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from .hierarchy import HierarchySpec, LevelSpec, SparseOp

__all__ = ["split_even", "ShardedHierarchy", "DistMultilevelSolver", "DeviceOps", "shardable"]

SHARDABLE_SMOOTHERS = ("jacobi", "polynomial", "block_jacobi", "none")     # row-independent sweeps: each row reads the OLD iterate


def split_even(n: int, parts: int) -> np.ndarray:
    """Contiguous row-block offsets [parts+1]."""
    return np.array([(n * r) // parts for r in range(parts + 1)], dtype=np.int64)


def shardable(spec: HierarchySpec) -> bool:
    """Row-independent smoothers on every level; block (BSR) operators are cut along BLOCK rows, so the level
    operators must carry square blocks and P / R blocks that fit them (SA on a BSR matrix: (3,3), (3,6), (6,6) ...)."""
    for i, L in enumerate(spec.levels[:-1]):
        bs = L.A.blocksize[0]
        if L.A.blocksize != (bs, bs) or L.A.shape[0] % bs:
            return False
        nxt = spec.levels[i + 1].A.blocksize[0] if spec.levels[i + 1].A.blocksize[0] == spec.levels[i + 1].A.blocksize[1] else 0
        if nxt == 0 or L.P.blocksize != (bs, nxt) or L.R.blocksize != (nxt, bs):
            return False
        for s in (L.pre, L.post):
            if s is None:
                continue
            if s.kind not in SHARDABLE_SMOOTHERS:
                return False
            if s.kind == "block_jacobi" and (bs == 1 or int(s.blocksize) != bs):
                return False
    return True


def _rows(op: SparseOp, r0: int, r1: int):
    """(block) rows [r0, r1): row pointer, (block) column ids, values (R*C per stored block, flat)"""
    p0, p1 = int(op.indptr[r0]), int(op.indptr[r1])
    per = int(op.blocksize[0]) * int(op.blocksize[1])
    return op.indptr[r0:r1 + 1] - op.indptr[r0], op.indices[p0:p1], np.ravel(op.data)[p0 * per:p1 * per]


def _ext_cols(op: SparseOp, r0: int, r1: int, c0: int, c1: int) -> np.ndarray:
    """Sorted unique global columns outside [c0,c1) touched by rows [r0,r1)."""
    cols = op.indices[int(op.indptr[r0]):int(op.indptr[r1])]
    ext = cols[(cols < c0) | (cols >= c1)]
    return np.unique(ext)


@dataclass
class LevelPlan:
    """Exchange plan of one level on one rank.  All counts and indices are in BLOCK units of the level (bs values per
    block; bs = 1 for scalar levels): ``*_s`` give the scalar sizes the vectors and messages have."""
    off: np.ndarray                     # [N+1] (block) row offsets of the level
    bs: int = 1
    n_owned: int = 0
    halo_cols: np.ndarray = None        # global ids of the halo entries, grouped by owner (ascending)
    recv: List[tuple] = field(default_factory=list)   # (src_rank, halo_begin, count)
    send: List[tuple] = field(default_factory=list)   # (dst_rank, send_begin, count)
    send_idx: np.ndarray = None         # owned-local indices to pack, grouped by destination

    @property
    def n_halo(self) -> int:
        return int(self.halo_cols.size)

    @property
    def n_local(self) -> int:
        return self.n_owned + self.n_halo

    @property
    def n_owned_s(self) -> int:
        return self.n_owned * self.bs

    @property
    def n_halo_s(self) -> int:
        return self.n_halo * self.bs

    @property
    def n_local_s(self) -> int:
        return self.n_local * self.bs

    def row0_s(self, rank) -> int:
        return int(self.off[rank]) * self.bs

    @property
    def send_idx_s(self) -> np.ndarray:
        """scalar indices to pack"""
        if self.bs == 1:
            return self.send_idx
        return (self.send_idx.astype(np.int64)[:, None] * self.bs + np.arange(self.bs)).ravel().astype(np.int32)


def _localize(op: SparseOp, r0: int, r1: int, c0: int, c1: int, halo_cols: np.ndarray) -> SparseOp:
    """(Block) rows [r0,r1) of ``op`` with (block) columns renumbered to [owned | halo]."""
    indptr, cols, data = _rows(op, r0, r1)
    owned = (cols >= c0) & (cols < c1)
    loc = np.empty(cols.size, dtype=np.int32)
    loc[owned] = (cols[owned] - c0).astype(np.int32)
    if (~owned).any():
        pos = np.searchsorted(halo_cols, cols[~owned])
        assert np.array_equal(halo_cols[pos], cols[~owned])
        loc[~owned] = ((c1 - c0) + pos).astype(np.int32)
    R, Cb = op.blocksize
    return SparseOp(op.fmt, ((r1 - r0) * R, ((c1 - c0) + int(halo_cols.size)) * Cb), (R, Cb),
                    np.ascontiguousarray(indptr, dtype=np.int32), loc, np.ascontiguousarray(data), op.src_format)


def _halo_needs(spec: HierarchySpec, offs, ns: int, world: int):
    """per level, per rank: sorted unique off-rank (block) columns of A_l (rows l), P_{l-1} (rows l-1), R_l (rows l+1)"""
    out = []
    for l in range(ns + 1):
        off = offs[l]
        needs = []
        for d in range(world):
            c0, c1 = int(off[d]), int(off[d + 1])
            parts = []
            if l < ns:
                parts.append(_ext_cols(spec.levels[l].A, c0, c1, c0, c1))
                ro = offs[l + 1]
                parts.append(_ext_cols(spec.levels[l].R, int(ro[d]), int(ro[d + 1]), c0, c1))
            if l > 0:
                fo = offs[l - 1]
                parts.append(_ext_cols(spec.levels[l - 1].P, int(fo[d]), int(fo[d + 1]), c0, c1))
            needs.append(np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int32))
        out.append(needs)
    return out


def _slim(sm):
    """a smoother spec without its per-row arrays (those are sliced per rank separately)"""
    import dataclasses
    return None if sm is None else dataclasses.replace(sm, Dinv=None)


class ShardedHierarchy:
    """Host-side partitioning of a HierarchySpec: what ONE rank needs -- its row blocks of the sharded levels with
    columns renumbered to [owned | halo], the exchange plans, its slices of the block-Jacobi inverses, the smoother
    parameters, and the collapsed coarse hierarchy.  Built from the full spec either by every rank for itself, or by
    rank 0 for everybody (``all_ranks`` + ``DistMultilevelSolver.from_rank0``): nothing here refers to the full spec once
    ``detach()`` has been called, so the object can be pickled to its rank."""

    @classmethod
    def all_ranks(cls, spec: HierarchySpec, world: int, min_rows: int = 200_000):
        """the parts of all ranks, one after another (the halo analysis is done once); each is detached from the spec"""
        shared = {}
        for d in range(world):
            part = cls(spec, d, world, min_rows, _shared=shared)
            part.detach()
            yield part

    def detach(self):
        self.spec = None
        return self

    def __init__(self, spec: HierarchySpec, rank: int, world: int, min_rows: int = 200_000, _shared=None):
        if not shardable(spec):
            raise NotImplementedError("hierarchy is not shardable (order-exact Gauss-Seidel, or block shapes that do not "
                                      "line up across levels): run replicas instead")
        self.spec, self.rank, self.world = spec, rank, world
        nlev = len(spec.levels)
        # sharded levels: 0 .. ns-1 ; level ns is the collapse level (full vectors on every rank)
        ns = 0
        while ns < nlev - 1 and spec.levels[ns].A.shape[0] >= max(min_rows, world):
            ns += 1
        if ns == 0:
            raise NotImplementedError("nothing to shard: fine level smaller than min_rows")
        self.ns = ns
        bss = [int(spec.levels[l].A.blocksize[0]) for l in range(ns + 1)]          # values per block, level by level
        offs = [split_even(spec.levels[l].A.shape[0] // bss[l], world) for l in range(ns + 1)]   # block rows
        self.plans: List[LevelPlan] = []
        self.A: List[SparseOp] = []
        self.P: List[SparseOp] = []
        self.R: List[SparseOp] = []
        # halo of level l = union of off-rank columns of A_l (rows l), P_{l-1} (rows l-1), R_l (rows l+1)
        if _shared is not None and "needs" in _shared:
            all_needs = _shared["needs"]
        else:
            all_needs = _halo_needs(spec, offs, ns, world)
            if _shared is not None:
                _shared["needs"] = all_needs
        self.smoothers = [(_slim(spec.levels[l].pre), _slim(spec.levels[l].post)) for l in range(ns)]
        self.shape0 = tuple(spec.levels[0].A.shape)
        self.dtype = spec.dtype
        self.nc = int(spec.levels[ns].A.shape[0])
        for l in range(ns + 1):
            off = offs[l]
            needs = all_needs[l]          # per rank d: sorted unique external columns of level l
            me = rank
            plan = LevelPlan(off=off, bs=bss[l], n_owned=int(off[me + 1] - off[me]), halo_cols=needs[me].astype(np.int64))
            owner = np.searchsorted(off, plan.halo_cols, side="right") - 1
            for s in range(world):
                cnt = int(np.count_nonzero(owner == s))
                if cnt:
                    beg = int(np.searchsorted(owner, s, side="left"))
                    plan.recv.append((s, beg, cnt))
            sidx, beg = [], 0
            for d in range(world):
                if d == me:
                    continue
                mine = needs[d][(needs[d] >= off[me]) & (needs[d] < off[me + 1])]
                if mine.size:
                    sidx.append((mine - off[me]).astype(np.int32))
                    plan.send.append((d, beg, int(mine.size)))
                    beg += int(mine.size)
            plan.send_idx = np.concatenate(sidx) if sidx else np.zeros(0, dtype=np.int32)
            self.plans.append(plan)
        for l in range(ns):
            o, oc = offs[l], offs[l + 1]
            r0, r1 = int(o[rank]), int(o[rank + 1])
            q0, q1 = int(oc[rank]), int(oc[rank + 1])
            L = spec.levels[l]
            self.A.append(_localize(L.A, r0, r1, r0, r1, self.plans[l].halo_cols))
            self.P.append(_localize(L.P, r0, r1, q0, q1, self.plans[l + 1].halo_cols))
            self.R.append(_localize(L.R, q0, q1, r0, r1, self.plans[l].halo_cols))
        # block Jacobi: this rank's slice of the inverted diagonal blocks
        self.Dinv: List[dict] = []
        for l in range(ns):
            r0, r1 = int(offs[l][rank]), int(offs[l][rank + 1])
            self.Dinv.append({k: np.ascontiguousarray(sm.Dinv[r0:r1]) for k, sm in (("pre", spec.levels[l].pre), ("post", spec.levels[l].post))
                              if sm is not None and sm.kind == "block_jacobi"})
        # remaining (collapsed) hierarchy, replicated on every rank
        self.coarse_spec = HierarchySpec(levels=[LevelSpec(A=L.A, P=L.P, R=L.R, pre=L.pre, post=L.post)
                                                 for L in spec.levels[ns:]],
                                         coarse_kind=spec.coarse_kind, coarse_op=spec.coarse_op,
                                         coarse_name=spec.coarse_name)


# ----------------------------------------------------------------------------------- local ops
class DeviceOps:
    """Local arithmetic on the GPU through the C ABI.  Vectors are torch CUDA tensors (torch is
    plumbing: device memory + the process group); kernels run on the legacy default stream,
    which is torch's current stream, so collectives and kernels are ordered."""

    def __init__(self, device_index: int, dtype=np.float64):
        import torch
        from . import _capi as capi
        self.torch, self.capi = torch, capi
        self.device = torch.device("cuda", device_index)
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        capi.check(capi.lib().pamg_set_device(device_index), "pamg_set_device")

    # buffers
    def vector(self, n):
        return self.torch.zeros(max(int(n), 1), dtype=self.tdtype, device=self.device)

    def index(self, idx):
        return self.torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(self.device)

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=self.dtype)).to(self.device)

    def to_host(self, t, n):
        return t[:n].cpu().numpy()

    def _p(self, t, offset=0):
        return C.c_void_p(t.data_ptr() + offset * t.element_size())

    # operators
    def matrix(self, op: SparseOp):
        from .multilevel import DeviceMatrix
        return DeviceMatrix(op)

    def spmv(self, M, mode, x, y, b=None, c=0.0):
        self.capi.check(self.capi.lib().pamg_matrix_spmv(M.handle, mode, self._p(x), self._p(b) if b is not None else None,
                                                         float(c), self._p(y), None), "spmv")

    def jacobi_step(self, M, x_in, b, x_out, omega):
        self.capi.check(self.capi.lib().pamg_matrix_jacobi_step(M.handle, self._p(x_in), self._p(b), self._p(x_out),
                                                                float(omega), None), "jacobi_step")

    def block_jacobi_step(self, M, Dinv, x_in, b, x_out, omega):
        self.capi.check(self.capi.lib().pamg_matrix_block_jacobi_step(M.handle, self._p(Dinv), self._p(x_in), self._p(b), self._p(x_out),
                                                                      float(omega), None), "block_jacobi_step")

    def resid_sumsq(self, M, x, b):
        out = self.torch.zeros(1, dtype=self.torch.float64, device=self.device)
        self.capi.check(self.capi.lib().pamg_matrix_resid_sumsq(M.handle, self._p(x), self._p(b), self._p(out), None),
                        "resid_sumsq")
        return out

    def axpy(self, n, a, x, y):
        self.capi.check(self.capi.lib().pamg_vec_axpy(self.capi.dtype_code(self.dtype), int(n), float(a), self._p(x),
                                                      self._p(y), None), "axpy")

    def scale(self, n, a, x, y):
        self.capi.check(self.capi.lib().pamg_vec_scale(self.capi.dtype_code(self.dtype), int(n), float(a), self._p(x),
                                                       self._p(y), None), "scale")

    def gather(self, n, idx, src, dst):
        self.capi.check(self.capi.lib().pamg_vec_gather(self.capi.dtype_code(self.dtype), int(n), self._p(idx),
                                                        self._p(src), self._p(dst), None), "gather")

    def coarse_solver(self, spec: HierarchySpec):
        from .multilevel import DeviceMultilevelSolver
        return DeviceMultilevelSolver(spec, graph=True)

    def coarse_cycle(self, solver, x, b, cycle):
        self.capi.sync()                 # inputs were produced on the default stream
        self.capi.check(self.capi.lib().pamg_solver_cycle(solver.handle, self._p(x), self._p(b), self.capi.CYCLE[cycle],
                                                          1, None), "coarse cycle")


# ------------------------------------------------------------------------------- the solver
class DistMultilevelSolver:
    """Sharded twin of ``DeviceMultilevelSolver`` (V-cycle, accel=None branch of ``solve``).

    ``group``: a ``torch.distributed`` process group (None = default).  All ranks call every
    method collectively with identical arguments (b, x0 are the GLOBAL vectors on the host;
    each rank uses its slice and ``solve`` returns the global solution on every rank).
    """

    def __init__(self, spec: Optional[HierarchySpec], ops=None, group=None, min_rows: int = 200_000, sharded=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._gloo = dist.get_backend(group) == "gloo"
        self.sh = sharded if sharded is not None else ShardedHierarchy(spec, self.rank, self.world, min_rows)
        if self.sh.rank != self.rank or self.sh.world != self.world:
            raise ValueError("sharded part belongs to another rank / world size")
        self.ops = ops if ops is not None else DeviceOps(int(__import__("os").environ.get("LOCAL_RANK", self.rank)),
                                                         self.sh.dtype)
        o = self.ops
        ns = self.sh.ns
        self.A = [o.matrix(m) for m in self.sh.A]
        self.P = [o.matrix(m) for m in self.sh.P]
        self.R = [o.matrix(m) for m in self.sh.R]
        self.send_idx = [o.index(p.send_idx_s) for p in self.sh.plans]
        self.send_buf = [o.vector(p.send_idx.size * p.bs) for p in self.sh.plans]
        nl = [p.n_local_s for p in self.sh.plans]
        self.Dinv = [{k: o.from_host(v.ravel()) for k, v in d.items()} for d in self.sh.Dinv]
        self.x = [o.vector(nl[l]) for l in range(ns + 1)]
        self.xalt = [o.vector(nl[l]) for l in range(ns)]
        self.b = [o.vector(nl[l]) for l in range(ns + 1)]
        self.r = [o.vector(nl[l]) for l in range(ns)]
        self.h = [[o.vector(nl[l]), o.vector(nl[l])] if self._has_poly(l) else None for l in range(ns)]
        nc = self.sh.nc
        self.bc_full = o.vector(nc)
        self.xc_full = o.vector(nc)
        cplan = self.sh.plans[ns]
        c0 = int(cplan.off[self.rank])
        fill = np.concatenate([np.arange(c0, c0 + cplan.n_owned, dtype=np.int64), cplan.halo_cols])     # blocks: owned | halo
        self.c_fill_idx = o.index((fill[:, None] * cplan.bs + np.arange(cplan.bs)).ravel().astype(np.int32))
        self.coarse = o.coarse_solver(self.sh.coarse_spec)
        self.shape = self.sh.shape0

    @classmethod
    def from_rank0(cls, spec: Optional[HierarchySpec], ops=None, group=None, min_rows: int = 200_000):
        """The hierarchy exists on rank 0 only (``spec`` is None elsewhere): rank 0 partitions it for everybody and
        scatters the parts, so no other rank ever builds or holds the full hierarchy (a 512^3 SA hierarchy is 78 GB on the
        host).  Collective over ``group``."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        parts = list(ShardedHierarchy.all_ranks(spec, world, min_rows)) if rank == 0 else None
        mine = [None]
        dist.scatter_object_list(mine, parts, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        return cls(None, ops=ops, group=group, min_rows=min_rows, sharded=mine[0])

    def _has_poly(self, l):
        return any(s is not None and s.kind == "polynomial" for s in self.sh.smoothers[l])

    # ---- communication
    def _staged(self, t):
        """gloo cannot move device memory point-to-point: stage through the host then (single-GPU
        test rigs; RCCL moves device buffers directly)."""
        return self._gloo and getattr(t, "is_cuda", False)

    def _all_reduce(self, t):
        if self._staged(t):
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)

    def exchange(self, l, v):
        """Fill the halo part of level-l vector ``v`` from its owners."""
        plan = self.sh.plans[l]
        if not plan.send and not plan.recv:
            return
        dist = self.dist
        bs, no = plan.bs, plan.n_owned_s
        if plan.send_idx.size:
            self.ops.gather(plan.send_idx.size * bs, self.send_idx[l], v, self.send_buf[l])
        if self._staged(v):
            sb = self.send_buf[l].cpu()
            rb = sb.new_zeros(max(plan.n_halo_s, 1))
            reqs = [dist.P2POp(dist.irecv, rb[beg * bs:(beg + cnt) * bs], self._peer(src), self.group) for (src, beg, cnt) in plan.recv]
            reqs += [dist.P2POp(dist.isend, sb[beg * bs:(beg + cnt) * bs], self._peer(dst), self.group) for (dst, beg, cnt) in plan.send]
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
            if plan.n_halo:
                v[no:no + plan.n_halo_s].copy_(rb[:plan.n_halo_s])
            return
        reqs = []
        for (src, beg, cnt) in plan.recv:
            reqs.append(dist.P2POp(dist.irecv, v[no + beg * bs: no + (beg + cnt) * bs], self._peer(src), self.group))
        for (dst, beg, cnt) in plan.send:
            reqs.append(dist.P2POp(dist.isend, self.send_buf[l][beg * bs: (beg + cnt) * bs], self._peer(dst), self.group))
        for w in dist.batch_isend_irecv(reqs):
            w.wait()

    def _peer(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    # ---- smoothers (reference: relaxation.py:349-420, 585-659)
    def _smooth(self, l, s, x_zero):
        if s is None or s.kind == "none":
            return
        o, A = self.ops, self.A[l]
        n = self.sh.plans[l].n_owned_s
        if s.kind == "block_jacobi":
            Dinv = self.Dinv[l]["pre" if s is self.sh.smoothers[l][0] else "post"]
            for it in range(s.iterations):
                if not (x_zero and it == 0):
                    self.exchange(l, self.x[l])
                o.block_jacobi_step(A, Dinv, self.x[l], self.b[l], self.xalt[l], s.omega)
                self.x[l], self.xalt[l] = self.xalt[l], self.x[l]
            return
        if s.kind == "jacobi":
            for it in range(s.iterations):
                if not (x_zero and it == 0):             # halo of an all-zero iterate is zero already
                    self.exchange(l, self.x[l])
                o.jacobi_step(A, self.x[l], self.b[l], self.xalt[l], s.omega)
                self.x[l], self.xalt[l] = self.xalt[l], self.x[l]
            return
        co = np.asarray(s.coefficients, dtype=np.float64)
        for it in range(s.iterations):
            if x_zero and it == 0:
                res = self.b[l]
            else:
                self.exchange(l, self.x[l])
                o.spmv(A, 2, self.x[l], self.r[l], b=self.b[l])          # res = b - A x
                res = self.r[l]
            if co.size == 1:
                o.axpy(n, co[0], res, self.x[l])
                continue
            hc, hn = self.h[l]
            o.scale(n, co[0], res, hc)
            for k in range(1, co.size - 1):
                self.exchange(l, hc)
                o.spmv(A, 3, hc, hn, b=res, c=co[k])                     # h = c*res + A h
                hc, hn = hn, hc
            self.exchange(l, hc)
            o.spmv(A, 4, hc, self.x[l], b=res, c=co[-1])                 # x += c*res + A h

    # ---- one V-cycle on the sharded levels (multilevel.py:584-662)
    def _cycle(self, l, x_zero, cycle="V"):
        o, sh = self.ops, self.sh
        pre, post = self.sh.smoothers[l]
        self._smooth(l, pre, x_zero)
        self.exchange(l, self.x[l])
        o.spmv(self.A[l], 2, self.x[l], self.r[l], b=self.b[l])          # r = b - A x
        self.exchange(l, self.r[l])
        if l + 1 < sh.ns:
            o.spmv(self.R[l], 0, self.r[l], self.b[l + 1])               # b_c = R r
            self.x[l + 1].zero_()
            self._cycle(l + 1, True, cycle)
            self.exchange(l + 1, self.x[l + 1])
            o.spmv(self.P[l], 1, self.x[l + 1], self.x[l])               # x += P x_c
        else:
            cplan = sh.plans[sh.ns]
            c0 = cplan.row0_s(self.rank)
            self.bc_full.zero_()
            o.spmv(self.R[l], 0, self.r[l], self.b[sh.ns])               # owned slice of b_c
            self.bc_full[c0: c0 + cplan.n_owned_s].copy_(self.b[sh.ns][:cplan.n_owned_s])
            self._all_reduce(self.bc_full)                               # disjoint slices -> full b_c everywhere
            self.xc_full.zero_()
            o.coarse_cycle(self.coarse, self.xc_full, self.bc_full, cycle)
            o.gather(cplan.n_local_s, self.c_fill_idx, self.xc_full, self.x[sh.ns])
            o.spmv(self.P[l], 1, self.x[sh.ns], self.x[l])               # x += P x_c
        self._smooth(l, post, False)

    def resid_norm(self):
        self.exchange(0, self.x[0])
        ss = self.ops.resid_sumsq(self.A[0], self.x[0], self.b[0])
        self._all_reduce(ss)
        return float(np.sqrt(float(ss.item())))

    def load(self, b, x0):
        p = self.sh.plans[0]
        r0, no = p.row0_s(self.rank), p.n_owned_s
        self.b[0][:no].copy_(self.ops.from_host(np.ravel(b)[r0:r0 + no]))
        self.x[0][:no].copy_(self.ops.from_host(np.ravel(x0)[r0:r0 + no]))

    def iterate(self, k, cycle="V", want_residuals=True):
        """k x (V-cycle + convergence-check norm) on the resident sharded state."""
        out = []
        for _ in range(k):
            self._cycle(0, False, cycle)
            if want_residuals:
                out.append(self.resid_norm())
        return out

    def gather_solution(self):
        p = self.sh.plans[0]
        full = self.ops.vector(self.shape[0])
        full.zero_()
        r0 = p.row0_s(self.rank)
        full[r0:r0 + p.n_owned_s].copy_(self.x[0][:p.n_owned_s])
        self._all_reduce(full)
        return self.ops.to_host(full, self.shape[0])

    def solve(self, b, x0=None, tol=1e-5, maxiter=100, cycle="V", residuals=None, return_info=False):
        """accel=None branch of MultilevelSolver.solve (multilevel.py:537-582), V-cycle."""
        if str(cycle).upper() != "V":
            raise NotImplementedError("sharded path: V-cycle only")
        b = np.asarray(b)
        x = np.zeros_like(b) if x0 is None else np.array(x0)
        self.load(b, x)
        normb = float(np.linalg.norm(b))
        normb = 1.0 if normb == 0.0 else normb
        hist = [self.resid_norm()]
        it, info = 0, 0
        while True:
            self._cycle(0, False, "V")
            it += 1
            nr = self.resid_norm()
            hist.append(nr)
            if nr < tol * normb:
                info = 0
                break
            if it == maxiter:
                info = it
                break
        if residuals is not None:
            residuals[:] = hist
        xs = self.gather_solution()
        return (xs, info) if return_info else xs
