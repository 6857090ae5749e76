"""Multi-process tests of the row-sharded cycle (pyamg_amd/dist.py).

CPU (gloo, world_size 2 and 3): partition / halo plans / exchange / collapse logic with the
oracle doing the local arithmetic -- the sharded iterates must be BIT-IDENTICAL to the
unsharded reference iterates (per-row arithmetic is unchanged by sharding).
GPU (marked gpu): the same with the real HIP kernels, two ranks sharing the box's one GPU and
gloo as transport (RCCL needs one GPU per rank; the driver exercises that path)."""
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, name, backend, min_rows, tmp_path):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(r), str(world), str(port),
                               name, backend, str(min_rows), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(tmp_path / f"out_{r}.npz") for r in range(world)]


def test_partition_and_halo_plans():
    from pyamg_amd.dist import ShardedHierarchy, split_even
    from pyamg_amd.hierarchy import load_spec
    from conftest import GOLDEN
    spec, _ = load_spec(GOLDEN / "hier_sa2d_jacobi.npz")
    assert list(split_even(10, 3)) == [0, 3, 6, 10]
    world = 3
    shs = [ShardedHierarchy(spec, r, world, min_rows=100) for r in range(world)]
    ns = shs[0].ns
    assert ns >= 2
    for l in range(ns + 1):
        # every rank's sends match the peers' receives
        for r in range(world):
            for (dst, beg, cnt) in shs[r].plans[l].send:
                match = [c for (s, b, c) in shs[dst].plans[l].recv if s == r]
                assert match == [cnt]
                gl = shs[r].plans[l].send_idx[beg:beg + cnt] + shs[r].plans[l].off[r]
                (s, b, c), = [t for t in shs[dst].plans[l].recv if t[0] == r]
                assert np.array_equal(gl, shs[dst].plans[l].halo_cols[b:b + c])
    # local operators reproduce the global rows
    A = spec.levels[0].A.to_scipy()
    x = np.random.RandomState(0).rand(A.shape[1])
    y = A @ x
    for r in range(world):
        p = shs[r].plans[0]
        r0 = int(p.off[r])
        xl = np.concatenate([x[r0:r0 + p.n_owned], x[p.halo_cols]])
        assert np.array_equal(shs[r].A[0].to_scipy() @ xl, y[r0:r0 + p.n_owned]) or \
            np.allclose(shs[r].A[0].to_scipy() @ xl, y[r0:r0 + p.n_owned], rtol=0, atol=1e-12)


def test_gs_hierarchy_is_not_shardable():
    from pyamg_amd.dist import ShardedHierarchy, shardable
    from pyamg_amd.hierarchy import load_spec
    from conftest import GOLDEN
    spec, _ = load_spec(GOLDEN / "hier_sa2d_gs.npz")
    assert not shardable(spec)
    with pytest.raises(NotImplementedError):
        ShardedHierarchy(spec, 0, 2, min_rows=100)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", ["sa2d_jacobi", "sa2d_cheby", "rs2d_jacobi"])
def test_sharded_cycle_gloo_oracle(tmp_path, load_hier, world, name):
    outs = _run(world, name, "oracle", 100, tmp_path)
    spec, ex = load_hier(name)
    assert int(outs[0]["ns"]) >= 2 and outs[0]["halo"].max() > 0
    for o in outs:
        assert np.array_equal(o["x"], ex["x"])                    # bit-identical iterates
        assert np.max(np.abs(o["res"] - ex["res"])) <= 1e-12 * ex["res"][0]


@pytest.mark.parametrize("name,world", [("sa2d_cheby", 3), ("el3d_blockjacobi", 2)])
def test_hierarchy_built_on_rank0_and_scattered(tmp_path, load_hier, name, world):
    """DistMultilevelSolver.from_rank0: only rank 0 holds the hierarchy, partitions it for everybody (one halo analysis)
    and scatters the parts; same bits as when every rank partitions the full hierarchy itself"""
    outs = _run(world, name, "oracle+rank0", 100, tmp_path)
    spec, ex = load_hier(name)
    for o in outs:
        assert np.array_equal(o["x"], ex["x"])
        assert np.max(np.abs(o["res"] - ex["res"])) <= 1e-12 * ex["res"][0]


def test_all_ranks_parts_equal_per_rank_construction():
    import pickle
    from pyamg_amd.dist import ShardedHierarchy
    from pyamg_amd.hierarchy import load_spec
    from conftest import GOLDEN
    spec, _ = load_spec(GOLDEN / "hier_sa2d_jacobi.npz")
    world = 3
    parts = list(ShardedHierarchy.all_ranks(spec, world, min_rows=100))
    for r, part in enumerate(parts):
        own = ShardedHierarchy(spec, r, world, min_rows=100)
        assert part.spec is None and part.rank == r and part.ns == own.ns
        part = pickle.loads(pickle.dumps(part))                      # what scatter_object_list does to it
        for l in range(own.ns + 1):
            a, b = part.plans[l], own.plans[l]
            assert a.send == b.send and a.recv == b.recv and np.array_equal(a.halo_cols, b.halo_cols)
            assert np.array_equal(a.send_idx, b.send_idx)
        for l in range(own.ns):
            for x, y in ((part.A[l], own.A[l]), (part.P[l], own.P[l]), (part.R[l], own.R[l])):
                assert np.array_equal(x.indptr, y.indptr) and np.array_equal(x.indices, y.indices) and np.array_equal(x.data, y.data)
            assert part.smoothers[l][0].kind == spec.levels[l].pre.kind and part.smoothers[l][0].Dinv is None


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_block_operators_gloo_oracle(tmp_path, load_hier, world):
    """3-D elasticity, BSR (3,3) -> (6,6) levels with (3,6) / (6,3) transfer blocks, block Jacobi: the hierarchy is cut
    along BLOCK rows, halos travel as whole blocks; iterates bit-identical to the unsharded reference run"""
    from pyamg_amd.dist import shardable
    name = "el3d_blockjacobi"
    spec, ex = load_hier(name)
    assert shardable(spec) and spec.levels[0].A.blocksize == (3, 3)
    outs = _run(world, name, "oracle", 100, tmp_path)
    assert int(outs[0]["ns"]) >= 1 and outs[0]["halo"].max() > 0
    for o in outs:
        assert np.array_equal(o["x"], ex["x"])
        assert np.max(np.abs(o["res"] - ex["res"])) <= 1e-12 * ex["res"][0]


def test_part_travels_as_arrays():
    """from_rank0 ships a part as a small pickled skeleton + its arrays one by one: nothing sizeable is left in the
    skeleton and the part comes back identical"""
    from pyamg_amd.dist import ShardedHierarchy, _join_part, _split_part
    from pyamg_amd.hierarchy import load_spec
    from conftest import GOLDEN
    spec, _ = load_spec(GOLDEN / "hier_el3d_blockjacobi.npz")
    part = list(ShardedHierarchy.all_ranks(spec, 2, min_rows=100))[1]
    skeleton, arrays = _split_part(part)
    assert len(skeleton) < 64 * 1024 and sum(a.nbytes for a in arrays) > 10 * len(skeleton)
    back = _join_part(skeleton, arrays)
    assert back.rank == 1 and back.ns == part.ns and back.nc == part.nc
    for l in range(part.ns):
        for x, y in ((back.A[l], part.A[l]), (back.P[l], part.P[l]), (back.R[l], part.R[l])):
            assert x.shape == y.shape and x.blocksize == y.blocksize
            assert np.array_equal(x.indptr, y.indptr) and np.array_equal(x.indices, y.indices) and np.array_equal(x.data, y.data)
        assert all(np.array_equal(back.Dinv[l][k], part.Dinv[l][k]) for k in part.Dinv[l])
        assert np.array_equal(back.plans[l].send_idx, part.plans[l].send_idx) and back.plans[l].recv == part.plans[l].recv


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["device", "devicepy", "device+rank0", "device+allgather"])
@pytest.mark.parametrize("name", ["sa2d_jacobi", "sa2d_cheby", "el3d_blockjacobi"])
def test_sharded_cycle_device_kernels(tmp_path, load_hier, name, backend):
    """two ranks (sharing the box's GPU, gloo transport) against the UNSHARDED device run of the same cycles: the
    per-row arithmetic does not change with the partition, so the iterates are bit-identical; and both stay within the
    usual distance of the reference's history.  "device" = the C++ driver (pamg_dist_*: interior rows overlapped with the
    halo exchange), "devicepy" = the Python schedule it mirrors, "+rank0" = hierarchy shipped from rank 0 as arrays."""
    from pyamg_amd import DeviceMultilevelSolver
    if backend == "device+rank0" and name != "sa2d_cheby":
        pytest.skip("one hierarchy is enough for the shipping path")
    # "+allgather": the exchange as ONE all-gather of the owned parts per operator application (SURVEY 8e's fallback and
    # correctness baseline; formed by the all-reduce callback on this rig) -- the same halo values, so the same bits
    outs = _run(2, name, backend, 100, tmp_path)
    if backend.startswith("device") and backend != "devicepy":
        assert all(int(o["exchanges"]) > 0 for o in outs)
        if name != "el3d_blockjacobi":          # scalar shards: every exchange runs its interior ranges meanwhile
            assert all(int(o["overlapped"]) > 0 for o in outs)
    spec, ex = load_hier(name)
    dml = DeviceMultilevelSolver(spec)
    r1 = []
    x1 = dml.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=int(ex["k"]), residuals=r1)
    dml.free()
    for o in outs:
        assert np.array_equal(o["x"], x1)
        assert np.max(np.abs(o["res"] - ex["res"])) <= 1e-10 * ex["res"][0]
        assert np.linalg.norm(o["x"] - ex["x"]) <= 1e-12 * np.linalg.norm(ex["x"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sa2d_jacobi", "sa2d_cheby", "el3d_blockjacobi"])
def test_sharded_driver_one_rank_is_the_resident_cycle(load_hier, name):
    """one rank, no peers: the C++ driver replays its whole iteration from one hipGraph and must reproduce the resident
    engine bit for bit -- iterate AND residual norms (no all-reduce to reorder the sum)"""
    from pyamg_amd import DeviceMultilevelSolver
    from pyamg_amd.dist import DeviceOps, DistMultilevelSolver
    spec, ex = load_hier(name)
    k = int(ex["k"])
    dml = DeviceMultilevelSolver(spec)
    r1 = []
    x1 = dml.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, residuals=r1)
    dml.free()
    sol = DistMultilevelSolver(spec, ops=DeviceOps(0, spec.dtype), min_rows=100)
    info = sol.native.info()
    assert info["transport"] == "none" and info["graph"] == 1 and info["sharded_levels"] == sol.sh.ns >= 1
    r2 = []
    x2 = sol.solve(ex["b"], x0=ex["x0"], tol=1e-30, maxiter=k, residuals=r2)
    assert np.array_equal(x2, x1)
    assert np.array_equal(np.asarray(r2), np.asarray(r1))
    # resident state API: k cycles in one call, norms only
    sol.load(ex["b"], ex["x0"])
    assert np.array_equal(np.asarray(sol.iterate(k)), np.asarray(r1)[1:])


@pytest.mark.gpu
def test_rccl_binds_and_initialises_a_communicator():
    """the production transport: librccl is bound at run time and a communicator of our own comes up (one rank here --
    RCCL refuses two ranks on one GPU; the N > 1 exchange itself runs on the multi-GPU node only).  In a process of its
    own with torch imported FIRST, like every multi-rank process of this package: torch ships its own HIP runtime and
    librccl, and the copy that is loaded first is the one the whole process has to use."""
    code = (
        "import sys, ctypes as C, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "from pyamg_amd import _capi as capi\n"
        "lib = capi.lib()\n"
        "ident = np.zeros(128, dtype=np.uint8)\n"
        "capi.check(lib.pamg_dist_rccl_unique_id(capi.ptr(ident)), 'unique_id')\n"
        "assert ident.any()\n"
        "h = C.c_void_p()\n"
        "capi.check(lib.pamg_dist_create(C.byref(h), capi.F64, 0, 1), 'create')\n"
        "capi.check(lib.pamg_dist_set_rccl(h, capi.ptr(ident)), 'set_rccl')\n"
        "info = (C.c_int64 * 8)()\n"
        "capi.check(lib.pamg_dist_info(h, info), 'info')\n"
        "assert info[1] == 2\n"
        "lib.pamg_dist_destroy(h)\n"
        "print('rccl communicator ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl communicator ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
def test_rccl_self_sendrecv():
    """Every RCCL entry point of the sharded cycle EXECUTED on the hardware through the dlsym'd table with this library's
    enum constants (pamg_rccl_selftest): a one-rank communicator, a grouped ncclSend + ncclRecv to itself on a comm stream
    ordered against the main stream by the events of the halo exchange, ncclAllGather, a one-element ncclAllReduce --
    every received float64 must be the value sent.  (Two ranks need two GPUs: the driver's node.)"""
    code = (
        "import sys, ctypes as C\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "from pyamg_amd import _capi as capi\n"
        "lib = capi.lib()\n"
        "assert lib.pamg_rccl_available() == 0\n"
        "for n in (1, 1000, 1 << 20):\n"
        "    e = C.c_double(-1.0)\n"
        "    capi.check(lib.pamg_rccl_selftest(n, C.byref(e)), 'pamg_rccl_selftest')\n"
        "    assert e.value == 0.0, (n, e.value)\n"
        "print('rccl self send/recv ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl self send/recv ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.gpu
def test_driver_buffers_as_torch_tensors():
    """the second production transport (torch.distributed point-to-point on the driver's own buffers) rests on presenting
    a raw device allocation of this library to torch without a copy: values written by us are seen through the tensor,
    values written through the tensor are seen by us, and an NCCL collective runs on it (one rank here)."""
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import torch, torch.distributed as dist\n"
        "from pyamg_amd import _capi as capi\n"
        "from pyamg_amd.dist import _device_tensor\n"
        "a = np.arange(1000, dtype=np.float64) * 0.5\n"
        "d = capi.DeviceArray.from_host(a)\n"
        "t = _device_tensor(d.ptr, a.size, np.float64, torch.device('cuda', 0))\n"
        "assert t.is_cuda and t.dtype == torch.float64 and np.array_equal(t.cpu().numpy(), a)\n"
        "t[10:20] = -3.0\n"
        "torch.cuda.synchronize()\n"
        "b = d.download()\n"
        "assert (b[10:20] == -3.0).all() and np.array_equal(b[:10], a[:10])\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "dist.all_reduce(t[:100])\n"
        "torch.cuda.synchronize()\n"
        "assert np.array_equal(d.download()[:10], a[:10])\n"
        "dist.destroy_process_group()\n"
        "print('tensor view ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tensor view ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
