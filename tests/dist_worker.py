"""Worker for the multi-process tests of pyamg_amd.dist (spawned by tests/test_dist.py).

``OracleOps`` is the CPU twin of ``pyamg_amd.dist.DeviceOps``: same interface, local arithmetic
done by the oracle on torch CPU tensors, so the partition / halo-exchange / cycle logic runs
under gloo without a GPU.  With ``backend == "device"`` the real DeviceOps is used instead
(ranks share the one GPU of the box; transport stays gloo via host staging)."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class OracleOps:
    def __init__(self, dtype=np.float64):
        import torch
        from oracle import oracle as orc
        self.torch, self.orc = torch, orc
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32

    def vector(self, n):
        return self.torch.zeros(max(int(n), 1), dtype=self.tdtype)

    def index(self, idx):
        return self.torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32))

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=self.dtype))

    def to_host(self, t, n):
        return t[:n].numpy().copy()

    def matrix(self, op):
        return op

    def _mv(self, M, x):
        return self.orc.matvec(M, x.numpy()[:M.shape[1]])

    def spmv(self, M, mode, x, y, b=None, c=0.0):
        n = M.shape[0]
        s = self._mv(M, x)
        yn = y.numpy()
        if mode == 0:
            yn[:n] = s
        elif mode == 1:
            yn[:n] = yn[:n] + s
        elif mode == 2:
            yn[:n] = b.numpy()[:n] - s
        elif mode == 3:
            yn[:n] = c * b.numpy()[:n] + s
        elif mode == 4:
            yn[:n] = yn[:n] + (c * b.numpy()[:n] + s)

    def jacobi_step(self, M, x_in, b, x_out, omega):
        # one sweep of the reference's jacobi on the local rows: temp = [owned | halo] values
        n = M.shape[0]
        xin = x_in.numpy()[:M.shape[1]].copy()
        xo = xin.copy()
        temp = xin.copy()                      # the kernel reads only temp; pre-filled incl. halo
        # orc.jacobi copies x->temp for swept rows itself and reads temp elsewhere
        if M.fmt == "csr":
            self.orc.jacobi(M.indptr, M.indices, M.data, xo, b.numpy()[:n].copy(), temp, 0, n, 1, self.dtype.type(omega))
        else:
            bs = M.blocksize[0]
            self.orc.bsr_jacobi(M.indptr, M.indices, M.data, xo, b.numpy()[:n].copy(), temp, 0, n // bs, 1, bs,
                                self.dtype.type(omega))
        x_out.numpy()[:n] = xo[:n]

    def block_jacobi_step(self, M, Dinv, x_in, b, x_out, omega):
        n, bs = M.shape[0], M.blocksize[0]
        xin = x_in.numpy()[:M.shape[1]].copy()
        xo, temp = xin.copy(), xin.copy()
        self.orc.block_jacobi(M.indptr, M.indices, M.data, xo, b.numpy()[:n].copy(), Dinv.numpy(), temp, 0, n // bs, 1,
                              self.dtype.type(omega), bs)
        x_out.numpy()[:n] = xo[:n]

    def resid_sumsq(self, M, x, b):
        n = M.shape[0]
        r = b.numpy()[:n] - self._mv(M, x)
        return self.torch.tensor([float(np.dot(r, r))], dtype=self.torch.float64)

    def axpy(self, n, a, x, y):
        y.numpy()[:n] = y.numpy()[:n] + self.dtype.type(a) * x.numpy()[:n]

    def scale(self, n, a, x, y):
        y.numpy()[:n] = self.dtype.type(a) * x.numpy()[:n]

    def gather(self, n, idx, src, dst):
        dst.numpy()[:n] = src.numpy()[idx.numpy()[:n]]

    def coarse_solver(self, spec):
        return self.orc.OracleSolver(spec)

    def coarse_cycle(self, solver, x, b, cycle):
        xn = np.zeros(solver.spec.levels[0].A.shape[0], dtype=self.dtype)
        bn = b.numpy()[:xn.size].copy()
        if len(solver.spec.levels) == 1:
            xn[:] = solver.coarse_solve(bn)
        else:
            solver.cycle(0, xn, bn, cycle)
        x.numpy()[:xn.size] = xn


def run(rank, world, port, name, backend, min_rows, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyamg_amd.hierarchy import load_spec
    from pyamg_amd.dist import DistMultilevelSolver
    spec, ex = load_spec(ROOT / "tests" / "golden" / f"hier_{name}.npz")
    scatter = "+rank0" in backend                 # the hierarchy exists on rank 0 only, parts are scattered
    exchange = "allgather" if "+allgather" in backend else "halo"    # the all-gather form of the C++ driver's exchange
    backend = backend.split("+")[0]
    native = None
    if backend in ("device", "devicepy"):
        from pyamg_amd.dist import DeviceOps
        ops = DeviceOps(0, spec.dtype)          # ranks share the box's one GPU; gloo traffic is staged through the host
        native = backend == "device"            # "device": the C++ driver (pamg_dist_*); "devicepy": the Python schedule, kernel by kernel
    else:
        ops = OracleOps(spec.dtype)
    if scatter:
        dtype = spec.dtype
        if rank != 0:
            spec = None                          # nothing but rank 0 ever sees the full hierarchy
        sol = DistMultilevelSolver.from_rank0(spec, ops=ops, min_rows=min_rows, native=native)
        assert sol.sh.spec is None and sol.sh.dtype == dtype
    else:
        sol = DistMultilevelSolver(spec, ops=ops, min_rows=min_rows, native=native, exchange=exchange)
    assert (sol.native is not None) == (backend == "device")
    if sol.native is not None:
        assert sol.native.info()["exchange"] == exchange
    k = int(ex["k"])
    res = []
    x = sol.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, residuals=res)
    info = sol.native.info() if sol.native is not None else {}
    np.savez(Path(out_dir) / f"out_{rank}.npz", x=x, res=np.array(res), ns=sol.sh.ns,
             halo=np.array([p.n_halo for p in sol.sh.plans]), exchanges=info.get("exchanges_per_iteration", -1),
             overlapped=info.get("overlapped_exchanges", -1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    rank, world, port, name, backend, min_rows, out_dir = sys.argv[1:8]
    run(int(rank), int(world), int(port), name, backend, int(min_rows), out_dir)
