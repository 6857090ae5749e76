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

Two executors of the same plan.  In production (``DeviceOps``) the cycle is driven from C++
(``csrc/pamg_dist.hip`` behind ``pamg_dist_*``, ``_NativeCycle`` below): pack -> exchange -> unpack are stream-ordered
work, the halo travels on a second stream (RCCL send/recv straight into the halo part of the vector) while the row
ranges that read owned columns only are already running, and with no peers (world = 1) the whole iteration is one
hipGraph.  The Python schedule of this module (``_cycle`` and friends, one call per operator through an ``ops``
object) is the specification of that driver and what the CPU tests run: they inject an oracle-backed twin of
``DeviceOps`` so the partition / halo / cycle logic is exercised under ``gloo`` without a GPU.
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
                                         coarse_name=spec.coarse_name, coarse_smoother=spec.coarse_smoother)


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
        M = DeviceMatrix(op)
        if __import__("os").environ.get("PAMG_AUTOTUNE", "1") != "0":
            M.autotune(allow_cap=True)          # speed only (LDS window / streaming flags of large operators); no order-exact sweeps here
        return M

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

    def __init__(self, spec: Optional[HierarchySpec], ops=None, group=None, min_rows: int = 200_000, sharded=None, native=None,
                 exchange: Optional[str] = None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        # halo exchange of the C++ driver: 'halo' = point to point with the actual neighbours (default), 'allgather' = every
        # rank's owned part gathered everywhere (the general fallback / correctness baseline of SURVEY 8e)
        self.exchange_mode = exchange or __import__("os").environ.get("PAMG_DIST_EXCHANGE", "halo")
        self.transport_tried = []
        if dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            self._gloo = dist.get_backend(group) == "gloo"
        else:                                    # a single process without a process group: one rank, nobody to talk to
            self.rank, self.world, self._gloo = 0, 1, False
        self.sh = sharded if sharded is not None else ShardedHierarchy(spec, self.rank, self.world, min_rows)
        if self.sh.rank != self.rank or self.sh.world != self.world:
            raise ValueError("sharded part belongs to another rank / world size")
        self.ops = ops if ops is not None else DeviceOps(int(__import__("os").environ.get("LOCAL_RANK", self.rank)),
                                                         self.sh.dtype)
        o = self.ops
        ns = self.sh.ns
        # the C++ driver runs the cycle whenever the local arithmetic is the HIP engine's (PAMG_DIST_NATIVE=0: the Python
        # schedule below, kernel by kernel -- the specification the driver is checked against)
        if native is None:
            native = isinstance(o, DeviceOps) and __import__("os").environ.get("PAMG_DIST_NATIVE", "1") != "0"
        self.A = [o.matrix(m) for m in self.sh.A]
        self.P = [o.matrix(m) for m in self.sh.P]
        self.R = [o.matrix(m) for m in self.sh.R]
        self.coarse = o.coarse_solver(self.sh.coarse_spec)
        self.shape = self.sh.shape0
        self.native = None
        if native:
            # every level vector, the exchange buffers and the collapse buffers live inside the driver
            self.native = self._native_with_checked_transport()
            return
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

    def _all_agree(self, ok: bool) -> bool:
        """logical AND of a flag over the ranks of the group"""
        if self.world == 1:
            return ok
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if not self._gloo:
            t = t.to(self.ops.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.cpu()[0]))

    def _native_with_checked_transport(self):
        """The C++ driver with its production transport (RCCL under an NCCL process group), CHECKED before the first
        cycle: every sharded level exchanges a vector of global indices and every received halo value is compared.  If the
        binding, the communicator or a single value fails on any rank, all ranks switch together to the second transport --
        torch.distributed's own point-to-point calls on the driver's device buffers -- and say so; the arithmetic is the
        same either way.  PAMG_DIST_TRANSPORT=rccl|torch pins the choice."""
        import os
        import sys
        want = os.environ.get("PAMG_DIST_TRANSPORT", "")
        if self.world == 1:
            return _NativeCycle(self, "none")
        if self._gloo:                                   # test rigs: several ranks on one GPU, staged through the host
            nat = _NativeCycle(self, "host")
            bad = nat.verify_exchange()
            okh = self._all_agree(not bad)
            self.transport_tried.append({"transport": "host callbacks (gloo)", "ok": bool(okh), "why": bad})
            if not okh:
                raise RuntimeError(f"rank {self.rank}: halo exchange self-test failed ({bad or 'on another rank'})")
            return nat
        nat, why = None, ""
        # agree on the binding BEFORE anybody enters a collective of RCCL's (a rank that cannot bind librccl would otherwise go
        # straight to the vote below while the others wait for it inside ncclCommInitRank)
        from . import _capi as capi_
        have = self._all_agree(capi_.lib().pamg_rccl_available() == 0)
        if want != "torch" and not have:
            self.transport_tried.append({"transport": "rccl", "ok": False, "why": "librccl could not be bound on every rank"})
            if want == "rccl":
                raise RuntimeError(f"rank {self.rank}: RCCL requested but librccl cannot be bound on every rank")
        if want != "torch" and have:
            try:
                nat = _NativeCycle(self, "rccl")
                bad = nat.verify_exchange()
                if bad:
                    why = f"halo self-test: {bad}"
            except Exception as e:                  # noqa: BLE001 -- any failure here means: use the other transport
                why = repr(e)
            ok = self._all_agree(nat is not None and not why)
            self.transport_tried.append({"transport": "rccl", "ok": bool(ok), "why": why or ("" if ok else "failed on another rank")})
            if ok or want == "rccl":
                if not ok:
                    raise RuntimeError(f"rank {self.rank}: RCCL transport failed its self-test ({why or 'on another rank'})")
                return nat
            if nat is not None:
                nat.free()
            print(f"[pyamg_amd.dist] rank {self.rank}: RCCL transport not usable ({why or 'failed on another rank'}); "
                  "using torch.distributed point-to-point on the driver's buffers", file=sys.stderr, flush=True)
        nat = _NativeCycle(self, "torch")
        bad = nat.verify_exchange()
        okt = self._all_agree(not bad)
        self.transport_tried.append({"transport": "torch point-to-point", "ok": bool(okt), "why": bad})
        if not okt:
            raise RuntimeError(f"rank {self.rank}: halo exchange self-test failed on both transports ({bad or 'on another rank'})")
        return nat

    @classmethod
    def model_rank(cls, part: "ShardedHierarchy", ops):
        """ONE rank's share of a ``part.world``-rank solve on this process's device, with nobody on the wire (the C++ driver's MODEL
        transport): the launches of that rank -- pack, interior / boundary ranges, collapse, replicated tail -- for TIMING; the
        iterates are not a solve.  bench.py builds the modelled 1 -> 8 GPU curve from such runs + the exchange plans."""
        self = cls.__new__(cls)
        import torch.distributed as dist
        self.dist, self.group = dist, None
        self.exchange_mode, self.transport_tried = "halo", []
        self.rank, self.world, self._gloo = part.rank, part.world, False
        self.sh, self.ops = part, ops
        self.A = [ops.matrix(m) for m in part.A]
        self.P = [ops.matrix(m) for m in part.P]
        self.R = [ops.matrix(m) for m in part.R]
        self.coarse = ops.coarse_solver(part.coarse_spec)
        self.shape = part.shape0
        self.native = _NativeCycle(self, "model" if part.world > 1 else "none")
        return self

    @classmethod
    def from_rank0(cls, spec: Optional[HierarchySpec], ops=None, group=None, min_rows: int = 200_000, native=None):
        """The hierarchy exists on rank 0 only (``spec`` is None elsewhere): rank 0 partitions it for everybody, one part
        after another, and ships every part to its rank AS ARRAYS -- a small pickled skeleton plus one point-to-point
        transfer per index / value array (``_send_part`` / ``_recv_part``) -- so no other rank ever builds or holds the full
        hierarchy (a 512^3 SA hierarchy is 78 GB on the host) and nothing multi-GB goes through pickle.  Collective."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        mine = None
        if rank == 0:
            for part in ShardedHierarchy.all_ranks(spec, world, min_rows):
                if part.rank == 0:
                    mine = part
                else:
                    _send_part(part, part.rank, group)
                    del part
        else:
            mine = _recv_part(0, group)
        return cls(None, ops=ops, group=group, min_rows=min_rows, sharded=mine, native=native)

    def _has_poly(self, l):
        return any(s is not None and s.kind == "polynomial" for s in self.sh.smoothers[l])

    # ---- communication
    def _staged(self, t):
        """gloo cannot move device memory point-to-point: stage through the host then (single-GPU
        test rigs; RCCL moves device buffers directly)."""
        return self._gloo and getattr(t, "is_cuda", False)

    def _all_reduce(self, t):
        if self.world == 1:
            return
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
        if self.native is not None:
            return self.native.resid_norm()
        self.exchange(0, self.x[0])
        ss = self.ops.resid_sumsq(self.A[0], self.x[0], self.b[0])
        self._all_reduce(ss)
        return float(np.sqrt(float(ss.item())))

    def load(self, b, x0):
        p = self.sh.plans[0]
        r0, no = p.row0_s(self.rank), p.n_owned_s
        if self.native is not None:
            self.native.load(np.ravel(x0)[r0:r0 + no], np.ravel(b)[r0:r0 + no])
            return
        self.b[0][:no].copy_(self.ops.from_host(np.ravel(b)[r0:r0 + no]))
        self.x[0][:no].copy_(self.ops.from_host(np.ravel(x0)[r0:r0 + no]))

    def iterate(self, k, cycle="V", want_residuals=True):
        """k x (V-cycle + convergence-check norm) on the resident sharded state."""
        if self.native is not None:
            if str(cycle).upper() != "V":
                raise NotImplementedError("sharded path: V-cycle only")
            return self.native.iterate(k, want_residuals)
        out = []
        for _ in range(k):
            self._cycle(0, False, cycle)
            if want_residuals:
                out.append(self.resid_norm())
        return out

    def gather_solution(self):
        p = self.sh.plans[0]
        r0 = p.row0_s(self.rank)
        if self.native is not None:
            mine = self.native.store()
            if self.world == 1:
                return mine
            import torch
            full = torch.zeros(self.shape[0], dtype=torch.float64 if mine.dtype == np.float64 else torch.float32)
            full[r0:r0 + p.n_owned_s] = torch.from_numpy(mine)
            if not self._gloo:
                full = full.to(self.ops.device)
            self.dist.all_reduce(full, group=self.group)           # disjoint slices -> the whole vector on every rank
            return full.cpu().numpy()
        full = self.ops.vector(self.shape[0])
        full.zero_()
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
            if self.native is not None:
                nr = self.native.iterate(1, True)[0]
            else:
                self._cycle(0, False, "V")
                nr = self.resid_norm()
            it += 1
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


# ------------------------------------------------------------------------------- the C++ driver
class _DeviceBuffer:
    """a raw device allocation presented through ``__cuda_array_interface__`` (what torch.as_tensor accepts without a copy)"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _device_tensor(ptr, count, dtype, device):
    """a torch tensor over ``count`` values of ``dtype`` at device address ``ptr`` (no copy; the memory stays the caller's)"""
    import torch
    ptr = getattr(ptr, "value", ptr)                   # ctypes.c_void_p or a plain address
    return torch.as_tensor(_DeviceBuffer(ptr, count, np.dtype(dtype).str), device=device)


class _NativeCycle:
    """``pamg_dist_*`` (csrc/pamg_dist.hip) wired to one rank's plan: the operators are the DeviceMatrix shards the
    solver already holds, the exchange plans are handed over in scalar units, the transport is RCCL when the process
    group is NCCL, two ctypes callbacks staging through the host when it is gloo (test rigs with several ranks on one
    GPU), none when the world is one rank."""

    def __init__(self, sol: "DistMultilevelSolver", transport: str = "auto"):
        from . import _capi as capi
        self.capi, self.sol = capi, sol
        if transport == "auto":
            transport = "none" if sol.world == 1 else ("host" if sol._gloo else "rccl")
        self.transport = transport
        lib = capi.lib()
        sh, ns = sol.sh, sol.sh.ns
        h = C.c_void_p()
        capi.check(lib.pamg_dist_create(C.byref(h), capi.dtype_code(sh.dtype), sol.rank, sol.world), "pamg_dist_create")
        self.handle = h
        self._keep = []
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        for l in range(ns):
            p = sh.plans[l]
            sp_, so_ = i32([d for (d, _, _) in p.send]), i64([b * p.bs for (_, b, _) in p.send] + [p.send_idx.size * p.bs])
            rp_, ro_ = i32([s_ for (s_, _, _) in p.recv]), i64([b * p.bs for (_, b, _) in p.recv] + [p.n_halo_s])
            sidx = i32(p.send_idx_s)
            capi.check(lib.pamg_dist_add_level(h, sol.A[l].handle, sol.P[l].handle, sol.R[l].handle, p.n_owned_s, p.n_halo_s,
                                               sp_.size, capi.ptr(sp_), capi.ptr(so_), capi.ptr(sidx),
                                               rp_.size, capi.ptr(rp_), capi.ptr(ro_)), f"pamg_dist_add_level({l})")
            self._set_allgather(lib, h, l, p)
        cp = sh.plans[ns]
        c0 = int(cp.off[sol.rank])
        fill = np.concatenate([np.arange(c0, c0 + cp.n_owned, dtype=np.int64), cp.halo_cols])
        fill = i32((fill[:, None] * cp.bs + np.arange(cp.bs)).ravel())
        capi.check(lib.pamg_dist_set_collapse(h, sol.coarse.handle, sh.nc, cp.row0_s(sol.rank), cp.n_owned_s, cp.n_halo_s,
                                              capi.ptr(fill)), "pamg_dist_set_collapse")
        self._set_allgather(lib, h, ns, cp)
        for l in range(ns):
            for which, sm in enumerate(sh.smoothers[l]):
                kind = "none" if sm is None else sm.kind
                co = None if kind != "polynomial" else np.ascontiguousarray(sm.coefficients, dtype=np.float64)
                Dinv = None
                if kind == "block_jacobi":
                    Dinv = np.ascontiguousarray(sh.Dinv[l]["pre" if which == 0 else "post"], dtype=sh.dtype)
                capi.check(lib.pamg_dist_set_smoother(h, l, which, capi.SMOOTH[kind], int(sm.iterations) if sm else 0,
                                                      float(sm.omega) if sm else 1.0, capi.ptr(co), 0 if co is None else co.size,
                                                      capi.ptr(Dinv), int(sm.blocksize) if sm else 1),
                           f"pamg_dist_set_smoother({l}, {kind})")
        if sol.world > 1:
            {"host": self._host_transport, "rccl": self._rccl_transport, "torch": self._torch_transport, "model": self._model_transport}[transport](lib)
        self.exchange = "halo"
        if sol.world > 1 and getattr(sol, "exchange_mode", "halo") == "allgather" and transport in ("host", "rccl"):
            capi.check(lib.pamg_dist_set_exchange(h, 1), "pamg_dist_set_exchange")
            self.exchange = "allgather"
        capi.check(lib.pamg_dist_finalize(h), "pamg_dist_finalize")

    def _set_allgather(self, lib, h, l, p):
        """the all-gather form of level l's exchange (the general fallback and correctness baseline): where every
        halo value sits in the vector gathered from all ranks' owned parts, each padded to the largest one"""
        capi = self.capi
        off = np.asarray(p.off, dtype=np.int64)
        cnt = int(np.max(np.diff(off))) * p.bs if off.size > 1 else p.n_owned_s
        hc = np.asarray(p.halo_cols, dtype=np.int64)
        owner = np.searchsorted(off, hc, side="right") - 1
        src = ((owner * cnt + (hc - off[owner]) * p.bs)[:, None] + np.arange(p.bs)).ravel() if hc.size else np.zeros(0, dtype=np.int64)
        if src.size and int(src.max()) >= 2 ** 31:
            return                                           # beyond 32-bit positions: the point-to-point form only
        src = np.ascontiguousarray(src, dtype=np.int32)
        capi.check(lib.pamg_dist_set_allgather(h, l, max(cnt, 1), capi.ptr(src)), f"pamg_dist_set_allgather({l})")

    def set_exchange(self, mode: str):
        """'halo' (point to point with the actual neighbours) or 'allgather'; between iterations"""
        self.capi.check(self.capi.lib().pamg_dist_set_exchange(self.handle, 1 if mode == "allgather" else 0), "pamg_dist_set_exchange")
        self.exchange = mode

    def verify_exchange(self) -> str:
        """One halo exchange per sharded level with global indices as values (pamg_dist_exchange_test): '' when every
        received value is the index of the column it stands for, else a description of the first mismatch."""
        capi, sol = self.capi, self.sol
        sh, dt = sol.sh, np.dtype(sol.sh.dtype)
        for l in range(sh.ns):
            p = sh.plans[l]
            first = int(p.off[sol.rank]) * p.bs
            xo = (first + np.arange(p.n_owned_s, dtype=np.int64)).astype(dt) % dt.type(2 ** 20 if dt == np.float32 else 2 ** 40)
            want = ((np.asarray(p.halo_cols, dtype=np.int64)[:, None] * p.bs + np.arange(p.bs)).ravel()).astype(dt) \
                % dt.type(2 ** 20 if dt == np.float32 else 2 ** 40)
            got = np.full(max(p.n_halo_s, 1), -1.0, dtype=dt)
            xo = np.ascontiguousarray(xo if xo.size else np.zeros(1, dtype=dt))
            capi.check(capi.lib().pamg_dist_exchange_test(self.handle, l, capi.ptr(xo), capi.ptr(got)), "pamg_dist_exchange_test")
            if p.n_halo_s and not np.array_equal(got[:p.n_halo_s], want):
                k = int(np.flatnonzero(got[:p.n_halo_s] != want)[0])
                return f"level {l}: halo value {k} is {got[k]!r}, expected {want[k]!r} ({int((got[:p.n_halo_s] != want).sum())} of {p.n_halo_s} wrong)"
        return ""

    # second production transport: torch.distributed's own point-to-point calls, straight on the driver's device buffers
    # (the buffers are wrapped as tensors through __cuda_array_interface__; blocking, like the host callbacks)
    def _torch_transport(self, lib):
        import traceback
        import torch
        sol, capi = self.sol, self.capi
        dist, npdt = sol.dist, np.dtype(sol.sh.dtype)
        dev = sol.ops.device
        EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

        def wrap(ptr, count, dt):
            return _device_tensor(ptr, count, dt, dev)

        def exchange(_user, level, send_buf, send_count, halo, halo_count):
            try:
                plan = sol.sh.plans[level]
                bs = plan.bs
                st = wrap(send_buf, send_count, npdt) if send_count else None
                rt = wrap(halo, halo_count, npdt) if halo_count else None
                reqs = [dist.P2POp(dist.irecv, rt[beg * bs:(beg + cnt) * bs], sol._peer(src), sol.group) for (src, beg, cnt) in plan.recv]
                reqs += [dist.P2POp(dist.isend, st[beg * bs:(beg + cnt) * bs], sol._peer(dst), sol.group) for (dst, beg, cnt) in plan.send]
                if reqs:
                    for w in dist.batch_isend_irecv(reqs):
                        w.wait()
                    torch.cuda.synchronize(dev)
                return 0
            except Exception:                    # noqa: BLE001 -- an exception must not unwind through the C frames
                traceback.print_exc()
                return capi.E_COMM

        def allreduce(_user, buf, count, dtype):
            try:
                dt = np.dtype(np.float64 if dtype == capi.F64 else np.float32)
                t = wrap(buf, count, dt)
                dist.all_reduce(t, group=sol.group)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:                    # noqa: BLE001
                traceback.print_exc()
                return capi.E_COMM

        self._cb = (EX(exchange), AR(allreduce))
        capi.check(lib.pamg_dist_set_callbacks(self.handle, C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None),
                   "pamg_dist_set_callbacks")

    # RCCL: rank 0 draws the id, everybody learns it through the process group, the communicator is this library's own
    def _model_transport(self, lib):
        """nobody on the wire (``DistMultilevelSolver.model_rank``): the rank's own launches among ``world`` ranks, for timing only"""
        self.capi.check(lib.pamg_dist_set_model_transport(self.handle), "pamg_dist_set_model_transport")

    def level_info(self, level: int) -> dict:
        a = (C.c_int64 * 8)()
        self.capi.check(self.capi.lib().pamg_dist_level_info(self.handle, int(level), a), "pamg_dist_level_info")
        return dict(zip(("owned", "halo", "exchanges_per_iteration", "send_peers", "recv_peers", "max_sent_to_one_peer", "max_received_from_one_peer",
                         "sent_per_exchange"), [int(v) for v in a]))

    def set_options(self, use_graph=None, overlap=None):
        self.capi.check(self.capi.lib().pamg_dist_set_options(self.handle, -1 if use_graph is None else int(bool(use_graph)),
                                                              -1 if overlap is None else int(bool(overlap))), "pamg_dist_set_options")

    def _rccl_transport(self, lib):
        import torch
        sol, capi = self.sol, self.capi
        ident = np.zeros(128, dtype=np.uint8)
        if sol.rank == 0:
            capi.check(lib.pamg_dist_rccl_unique_id(capi.ptr(ident)), "pamg_dist_rccl_unique_id")
        t = torch.from_numpy(ident).to(sol.ops.device)
        sol.dist.broadcast(t, src=sol._peer(0), group=sol.group)
        ident = np.ascontiguousarray(t.cpu().numpy())
        capi.check(lib.pamg_dist_set_rccl(self.handle, capi.ptr(ident)), "pamg_dist_set_rccl")

    # gloo cannot move device memory: two blocking callbacks staging through the host
    def _host_transport(self, lib):
        import traceback
        import torch
        sol, capi = self.sol, self.capi
        dist, npdt = sol.dist, np.dtype(sol.sh.dtype)
        EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

        def exchange(_user, level, send_buf, send_count, halo, halo_count):
            try:
                plan = sol.sh.plans[level]
                bs = plan.bs
                sb = np.empty(max(int(send_count), 1), dtype=npdt)
                if send_count:
                    capi.check(lib.pamg_memcpy_d2h(capi.ptr(sb), send_buf, int(send_count) * npdt.itemsize, None), "d2h")
                rb = np.zeros(max(int(halo_count), 1), dtype=npdt)
                st, rt = torch.from_numpy(sb), torch.from_numpy(rb)
                reqs = [dist.P2POp(dist.irecv, rt[beg * bs:(beg + cnt) * bs], sol._peer(src), sol.group) for (src, beg, cnt) in plan.recv]
                reqs += [dist.P2POp(dist.isend, st[beg * bs:(beg + cnt) * bs], sol._peer(dst), sol.group) for (dst, beg, cnt) in plan.send]
                for w in dist.batch_isend_irecv(reqs):
                    w.wait()
                if halo_count:
                    capi.check(lib.pamg_memcpy_h2d(halo, capi.ptr(rb), int(halo_count) * npdt.itemsize, None), "h2d")
                return 0
            except Exception:                    # noqa: BLE001 -- an exception must not unwind through the C frames
                traceback.print_exc()
                return capi.E_COMM

        def allreduce(_user, buf, count, dtype):
            try:
                dt = np.dtype(np.float64 if dtype == capi.F64 else np.float32)
                hb = np.empty(int(count), dtype=dt)
                capi.check(lib.pamg_memcpy_d2h(capi.ptr(hb), buf, hb.nbytes, None), "d2h")
                t = torch.from_numpy(hb)
                dist.all_reduce(t, group=sol.group)
                capi.check(lib.pamg_memcpy_h2d(buf, capi.ptr(hb), hb.nbytes, None), "h2d")
                return 0
            except Exception:                    # noqa: BLE001
                traceback.print_exc()
                return capi.E_COMM

        self._cb = (EX(exchange), AR(allreduce))
        capi.check(lib.pamg_dist_set_callbacks(self.handle, C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None),
                   "pamg_dist_set_callbacks")

    def load(self, x_owned, b_owned):
        """this rank's slices of x0 and b (HOST arrays) into the driver's level-0 vectors"""
        capi, dt = self.capi, np.dtype(self.sol.sh.dtype)
        xd = capi.DeviceArray.from_host(np.ascontiguousarray(x_owned, dtype=dt))
        bd = capi.DeviceArray.from_host(np.ascontiguousarray(b_owned, dtype=dt))
        try:
            capi.check(capi.lib().pamg_dist_load(self.handle, xd.ptr, bd.ptr), "pamg_dist_load")
        finally:
            xd.free()
            bd.free()

    def store(self) -> np.ndarray:
        """this rank's slice of the iterate (HOST array)"""
        capi = self.capi
        xd = capi.DeviceArray(self.sol.sh.plans[0].n_owned_s, self.sol.sh.dtype)
        try:
            capi.check(capi.lib().pamg_dist_store(self.handle, xd.ptr), "pamg_dist_store")
            return xd.download()
        finally:
            xd.free()

    def iterate(self, k, want_residuals=True):
        res = np.zeros(max(int(k), 1), dtype=np.float64) if want_residuals else None
        self.capi.check(self.capi.lib().pamg_dist_iterate(self.handle, int(k), self.capi.ptr(res)), "pamg_dist_iterate")
        if not want_residuals:
            self.capi.check(self.capi.lib().pamg_dist_sync(self.handle), "pamg_dist_sync")
            return []
        return [float(v) for v in res[:k]]

    def resid_norm(self):
        v = C.c_double(0.0)
        self.capi.check(self.capi.lib().pamg_dist_resid_norm(self.handle, C.byref(v)), "pamg_dist_resid_norm")
        return float(v.value)

    def info(self) -> dict:
        a = (C.c_int64 * 8)()
        self.capi.check(self.capi.lib().pamg_dist_info(self.handle, a), "pamg_dist_info")
        keys = ("sharded_levels", "transport", "exchanges_per_iteration", "overlapped_exchanges", "graph", "vector_bytes",
                "values_sent_per_round", "interior_ranges_level0")
        d = dict(zip(keys, [int(v) for v in a]))
        d["transport"] = {"none": "none", "host": "host callbacks (gloo)", "rccl": "rccl",
                          "torch": "torch.distributed point-to-point on the driver's buffers"}[self.transport]
        d["exchange"] = self.exchange
        d["transport_tried"] = list(getattr(self.sol, "transport_tried", []))
        return d

    def free(self):
        if getattr(self, "handle", None):
            try:
                self.capi._lib.pamg_dist_destroy(self.handle)
            except Exception:       # pragma: no cover
                pass
            self.handle = None

    def __del__(self):
        self.free()


# ------------------------------------------------------------------------------- shipping a part to its rank
_CHUNK = 1 << 28            # bytes per point-to-point message


def _split_part(part):
    """(skeleton bytes, arrays): the part pickled with every sizeable ndarray taken out and replaced by its number"""
    import io
    import pickle
    arrays = []

    class Pk(pickle.Pickler):
        def persistent_id(self, obj):
            if isinstance(obj, np.ndarray) and obj.dtype.kind in "iuf" and obj.nbytes >= 1024:
                arrays.append(np.ascontiguousarray(obj))
                return ("ndarray", len(arrays) - 1)
            return None

    f = io.BytesIO()
    Pk(f, protocol=4).dump(part)
    return f.getvalue(), arrays


def _join_part(skeleton: bytes, arrays):
    import io
    import pickle

    class Up(pickle.Unpickler):
        def persistent_load(self, pid):
            return arrays[pid[1]]

    return Up(io.BytesIO(skeleton)).load()


def _p2p_bytes(buf: np.ndarray, peer: int, group, send: bool):
    """one flat uint8 array, point to point in chunks; NCCL moves device memory only: staged through a device buffer"""
    import torch
    import torch.distributed as dist
    nccl = dist.get_backend(group) == "nccl"
    peer = peer if group is None else dist.get_global_rank(group, peer)
    stage = torch.empty(min(_CHUNK, max(buf.size, 1)), dtype=torch.uint8, device="cuda") if nccl else None
    for o in range(0, buf.size, _CHUNK):
        piece = torch.from_numpy(buf[o:o + _CHUNK])
        if not nccl:
            (dist.send if send else dist.recv)(piece, peer, group=group)
        elif send:
            stage[:piece.numel()].copy_(piece)
            dist.send(stage[:piece.numel()], peer, group=group)
        else:
            dist.recv(stage[:piece.numel()], peer, group=group)
            piece.copy_(stage[:piece.numel()])


def _send_part(part, dst: int, group=None):
    skeleton, arrays = _split_part(part)
    head = np.array([len(skeleton), len(arrays)], dtype=np.int64)
    _p2p_bytes(head.view(np.uint8), dst, group, True)
    _p2p_bytes(np.frombuffer(skeleton, dtype=np.uint8).copy(), dst, group, True)
    desc = np.array([[ord(a.dtype.kind), a.dtype.itemsize, a.ndim, a.nbytes] + list(a.shape) + [0] * (4 - a.ndim) for a in arrays],
                    dtype=np.int64).reshape(-1, 8)
    _p2p_bytes(desc.view(np.uint8).reshape(-1), dst, group, True)
    for a in arrays:
        _p2p_bytes(a.view(np.uint8).reshape(-1), dst, group, True)


def _recv_part(src: int, group=None):
    head = np.zeros(2, dtype=np.int64)
    _p2p_bytes(head.view(np.uint8), src, group, False)
    skeleton = np.zeros(int(head[0]), dtype=np.uint8)
    _p2p_bytes(skeleton, src, group, False)
    desc = np.zeros((int(head[1]), 8), dtype=np.int64)
    _p2p_bytes(desc.view(np.uint8).reshape(-1), src, group, False)
    arrays = []
    for kind, itemsize, ndim, nbytes, *shape in desc:
        a = np.empty(tuple(int(v) for v in shape[:int(ndim)]), dtype=np.dtype(f"{chr(int(kind))}{int(itemsize)}"))
        assert a.nbytes == int(nbytes)
        _p2p_bytes(a.view(np.uint8).reshape(-1), src, group, False)
        arrays.append(a)
    return _join_part(skeleton.tobytes(), arrays)
