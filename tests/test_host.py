"""CPU: host-side logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import GOLDEN, ROOT, golden_hierarchies
from pyamg_amd import _capi as capi
from pyamg_amd import hierarchy as H


def test_library_loads_and_exports_every_declared_symbol():
    hdr = (ROOT / "include" / "pyamg_amd.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pamg_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 60
    lib = ctypes.CDLL(str(capi.LIB_PATH))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    lib.pamg_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.pamg_version()
    lib.pamg_status_string.restype = ctypes.c_char_p
    assert lib.pamg_status_string(-2) == b"not supported on the device path"


def test_gfx950_code_object_present():
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={capi.LIB_PATH}"], capture_output=True, text=True)
    blob = capi.LIB_PATH.read_bytes()
    assert b"gfx950" in blob and b"csr_stream_kernel" in blob
    assert out.returncode in (0, 1)


def test_product_fails_loudly_without_device():
    if capi.device_count() > 0:
        pytest.skip("a device is visible")
    with pytest.raises(capi.DeviceUnavailable):
        capi.lib()
    from pyamg_amd import DeviceMultilevelSolver
    spec, _ = H.load_spec(GOLDEN / "hier_sa2d_gs.npz")
    with pytest.raises(capi.DeviceUnavailable):
        DeviceMultilevelSolver(spec)
    from pyamg_amd import relaxation as grelax
    A = sp.eye_array(4, format="csr")
    with pytest.raises(capi.DeviceUnavailable):
        grelax.gauss_seidel(A, np.zeros(4), np.ones(4))


def test_strict_false_hands_unsupported_hierarchies_to_the_wrapped_solver():
    """an unsupported configuration (here: a complex operator, instantiate.yml:2-6) raises NotImplementedError
    by default and, with strict=False, is handed to the caller's OWN solver object with one warning -- same answers as that
    object gives, nothing of ours computes on the host (needs no device: the refusal happens while reading the hierarchy)."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not built")
    import pyamg
    from pyamg_amd import DeviceMultilevelSolver
    A = pyamg.gallery.poisson((12, 12), format="csr").astype(np.complex128)
    A = sp.csr_array(A + 0.05j * sp.eye_array(A.shape[0]))
    np.random.seed(0)
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10)
    with pytest.raises(NotImplementedError):
        DeviceMultilevelSolver(ml)
    with pytest.warns(RuntimeWarning, match="strict=False"):
        dml = DeviceMultilevelSolver(ml, strict=False)
    assert dml.fallback is ml and dml.shape == A.shape and "strict=False" in repr(dml)
    b = np.random.rand(A.shape[0]) + 1j * np.random.rand(A.shape[0])
    r1, r2 = [], []
    x1 = dml.solve(b, tol=1e-10, residuals=r1)
    x2 = ml.solve(b, tol=1e-10, residuals=r2)
    assert np.array_equal(x1, x2) and r1 == r2
    M = dml.aspreconditioner()
    assert np.array_equal(M @ b, ml.aspreconditioner() @ b)
    assert len(dml.levels) == len(ml.levels)


def test_product_never_imports_oracle():
    import pathlib
    for p in pathlib.Path(ROOT / "pyamg_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p
        assert "liboracle" not in src, p


def test_make_system_contract():
    """relaxation.py:15-97 / test_relaxation.py:64-111 (checked before any device use)."""
    from pyamg_amd.relaxation import make_system
    A = sp.eye_array(5, format="csr")
    x = np.zeros(5); b = np.ones(5)
    A2, x2, b2 = make_system(A, x.reshape(-1, 1), b, formats=["csr", "bsr"])
    assert x2.shape == (5,) and A2 is A
    with pytest.raises(ValueError):
        make_system(A, np.zeros(10)[::2], b)
    with pytest.raises(TypeError):
        make_system(A, x.astype(np.float32), b)
    with pytest.raises(ValueError):
        make_system(sp.csr_array(np.ones((2, 3))), np.zeros(2), np.zeros(2))
    with pytest.raises(ValueError):
        make_system(A, np.zeros(6), b)
    with pytest.raises(ValueError):
        make_system(A, [0.0] * 5, b)


def test_sparse_op_normalisation():
    rng = np.random.RandomState(5)
    M = sp.random(30, 20, density=0.2, random_state=rng, format="csr")
    for fmt in ("csr", "csc", "coo", "bsr"):
        op = H.sparse_op(M.asformat(fmt))
        assert op.indptr.dtype == np.int32 and op.indices.dtype == np.int32
        assert np.array_equal(op.to_scipy().toarray(), M.toarray())
    # csc -> csr with sorted columns: same per-row order as SciPy's column scatter
    op = H.sparse_op(M.tocsc())
    assert op.fmt == "csr" and op.src_format == "csc"
    for r in range(30):
        c = op.indices[op.indptr[r]:op.indptr[r + 1]]
        assert np.all(np.diff(c) > 0)
    B = sp.bsr_array(sp.kron(sp.eye_array(3), np.arange(6.0).reshape(2, 3) + 1))
    ob = H.sparse_op(B)
    assert ob.fmt == "bsr" and ob.blocksize == (2, 3) and ob.data.ndim == 1
    with pytest.raises(NotImplementedError):
        H.sparse_op(M.astype(np.complex128))


@pytest.mark.parametrize("name", golden_hierarchies())
def test_spec_roundtrip(tmp_path, load_hier, name):
    spec, ex = load_hier(name)
    H.save_spec(tmp_path / "s.npz", spec, **ex)
    spec2, ex2 = H.load_spec(tmp_path / "s.npz")
    assert len(spec2.levels) == len(spec.levels) and spec2.coarse_kind == spec.coarse_kind
    for a, b in zip(spec.levels, spec2.levels):
        assert np.array_equal(a.A.data, b.A.data) and a.A.fmt == b.A.fmt and a.A.blocksize == b.A.blocksize
        if a.pre is not None:
            assert a.pre.kind == b.pre.kind and a.pre.omega == b.pre.omega and a.pre.sweep == b.pre.sweep
    if spec.coarse_op is not None:
        assert np.array_equal(spec.coarse_op, spec2.coarse_op)
        assert np.isfortran(spec.coarse_op) == np.isfortran(spec2.coarse_op)


def test_extract_reads_back_smoother_parameters(tmp_path):
    """Smoother scalars come from the constructed solver (partial keywords / closure cells),
    never recomputed; dispatch on .func, not __name__ (8b)."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not built")
    import pyamg
    A = pyamg.gallery.poisson((30, 30), format="csr")
    np.random.seed(4)
    ml = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=("jacobi", {"omega": 4 / 3}),
                                           postsmoother=("chebyshev", {"degree": 2}))
    spec = H.extract(ml)
    for L, lv in zip(spec.levels[:-1], ml.levels[:-1]):
        assert L.pre.kind == "jacobi" and L.pre.omega == lv.presmoother.keywords["omega"]
        assert L.post.kind == "polynomial" and len(L.post.coefficients) == 2
    assert spec.levels[1].A.fmt == "bsr" and spec.levels[1].A.blocksize == (1, 1)
    assert spec.coarse_kind == "dense" and np.array_equal(spec.coarse_op, ml.coarse_solver.P)
    # block smoother names: blocksize 1 falls back to the point smoother under a block name
    ml2 = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother="block_gauss_seidel",
                                            postsmoother="block_jacobi")
    s2 = H.extract(ml2)
    assert s2.levels[0].pre.kind == "gauss_seidel" and s2.levels[0].post.kind == "jacobi"
    # Schwarz: the subdomains and inverted blocks are the arrays the reference's setup built (closure cells), and the
    # sweep's operator is lvl.Acsr -- shipped separately where the level's own storage differs (BSR(1,1) SA levels)
    ml3 = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother="schwarz", postsmoother=("schwarz", {"sweep": "backward"}))
    s3 = H.extract(ml3)
    for L, lv in zip(s3.levels[:-1], ml3.levels[:-1]):
        sub, sptr, inv, iptr = lv.Acsr.schwarz_parameters
        assert L.pre.kind == "schwarz" and L.post.sweep == "backward"
        assert np.array_equal(L.pre.subdomain, sub) and np.array_equal(L.pre.inv_subblock, inv)
        assert (L.pre.Ar is None) == (lv.A.format == "csr")
    ml3g = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=("gmres", {"maxiter": 2}), postsmoother="schwarz")
    # Krylov methods as smoothers (smoothing.py:794-830) and coarse solvers (multilevel.py:752-762): parameters from the closures
    s3g = H.extract(ml3g)
    assert s3g.levels[0].pre.kind == "gmres" and s3g.levels[0].pre.iterations == 2 and s3g.levels[0].pre.tol == 1e-12
    assert s3g.levels[0].pre.restart == 0 and s3g.levels[0].pre.At is None
    ml4 = pyamg.smoothed_aggregation_solver(A, max_coarse=10, coarse_solver="cg")
    s4 = H.extract(ml4)
    assert s4.coarse_kind == "relax" and s4.coarse_smoother.kind == "cg" and s4.coarse_smoother.iterations == 0
    assert s4.coarse_smoother.tol == 1e6 * np.finfo(np.float64).eps
    ml5 = pyamg.ruge_stuben_solver(A, max_coarse=10, presmoother=("cgnr", {"maxiter": 3, "tol": 1e-9}), postsmoother="cgne")
    s5 = H.extract(ml5)
    assert s5.levels[0].pre.kind == "cgnr" and s5.levels[0].pre.iterations == 3 and s5.levels[0].pre.tol == 1e-9
    At = s5.levels[0].post.At
    assert s5.levels[0].post.kind == "cgne" and At is not None and np.array_equal(At.indptr, A.T.tocsr().indptr)
    # the remaining Krylov coarse solvers and callables (multilevel.py:752-762, 786-788): solved on the host by the caller's own
    # solver object inside the device cycle; such a hierarchy holds a Python object and cannot be written to a file
    for cs in ("bicgstab", "cgs", "minres", lambda A_, b_: np.zeros_like(b_)):
        ml6 = pyamg.smoothed_aggregation_solver(A, max_coarse=10, coarse_solver=cs)
        s6 = H.extract(ml6)
        assert s6.coarse_kind == "host" and s6.coarse_host[0] is ml6.coarse_solver and s6.coarse_op is None
        with pytest.raises(NotImplementedError):
            H.save_spec(tmp_path / "host_coarse.npz", s6)
    # gmres's legacy restrt= is read as restart=; anything else the device loops do not restate is refused at extraction
    s7 = H.extract(pyamg.smoothed_aggregation_solver(A, max_coarse=10, coarse_solver=("gmres", {"restrt": 7})))
    assert s7.coarse_smoother.kind == "gmres" and s7.coarse_smoother.restart == 7
    with pytest.raises(NotImplementedError):
        H.extract(pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother=("cg", {"M": sp.eye_array(900, format="csr")})))


def test_extract_new_smoother_kinds_and_roundtrip(tmp_path):
    """AIR (no presmoother + FC Jacobi), the blackbox configuration (gauss_seidel_nr) and the other
    normal-equation smoothers: operands are built as the reference's wrappers build them, survive
    save/load, and what the reference itself cannot run (float32 Dinv mismatch) is refused."""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not built")
    import pyamg
    from pyamg.util.utils import get_diagonal
    m = 16
    Dx = sp.diags_array([np.ones(m), -np.ones(m - 1)], offsets=[0, -1], shape=(m, m))
    Dy = sp.diags_array([2 * np.ones(m), -np.ones(m - 1), -np.ones(m - 1)], offsets=[0, -1, 1], shape=(m, m))
    A = sp.csr_array(3.0 * sp.kron(sp.eye_array(m), Dx) + sp.kron(Dy, sp.eye_array(m)))
    A.sort_indices()
    np.random.seed(2)
    air = H.extract(pyamg.air_solver(A, max_coarse=20))
    L0 = air.levels[0]
    assert L0.pre.kind == "none" and L0.post.kind == "fc_jacobi" and L0.post.f_iterations == 2 and L0.post.c_iterations == 1
    assert L0.post.Fpts.dtype == np.int32 and len(L0.post.Fpts) + len(L0.post.Cpts) == A.shape[0]
    for kind, kw in (("gauss_seidel_nr", {"sweep": "symmetric", "iterations": 2}), ("gauss_seidel_ne", {"sweep": "backward"}),
                     ("jacobi_ne", {"iterations": 2, "omega": 0.5, "withrho": False})):
        np.random.seed(2)
        ml = pyamg.ruge_stuben_solver(A, max_coarse=10, presmoother=(kind, kw), postsmoother=(kind, kw))
        spec = H.extract(ml)
        s0 = spec.levels[0].pre
        assert s0.kind == kind and s0.iterations == kw.get("iterations", 1)
        Mc = sp.csc_array(A)
        if kind == "gauss_seidel_nr":
            assert np.array_equal(s0.Dinv, np.ravel(get_diagonal(Mc, norm_eq=1, inv=True))) and s0.sweep == "symmetric"
            assert np.array_equal(s0.At.indptr, Mc.indptr) and np.array_equal(s0.At.data, Mc.data)
        else:
            assert np.array_equal(s0.Dinv, np.ravel(get_diagonal(A, norm_eq=2, inv=True)))
        if kind == "jacobi_ne":
            assert s0.omega == 0.5 and np.array_equal(s0.At.data, 0.5 * Mc.data)
        H.save_spec(tmp_path / "s.npz", spec)
        spec2, _ = H.load_spec(tmp_path / "s.npz")
        t0 = spec2.levels[0].pre
        assert t0.kind == kind and np.array_equal(t0.Dinv, s0.Dinv) and (t0.At is None) == (s0.At is None)
        if s0.At is not None:
            assert np.array_equal(t0.At.data, s0.At.data) and np.array_equal(t0.At.indices, s0.At.indices)
        # every coarse RS level is CSR here; unsorted ones are shipped sorted for the NE kinds / get an Ar for NR
        for Ls, lv in zip(spec.levels[:-1], ml.levels[:-1]):
            if kind == "gauss_seidel_nr":
                assert (Ls.pre.Ar is None) == bool(lv.A.has_sorted_indices)
            else:
                M = sp.csr_array((Ls.A.data, Ls.A.indices, Ls.A.indptr), shape=Ls.A.shape)
                assert M.has_sorted_indices
    with pytest.raises(NotImplementedError):
        H._normal_equation_spec("gauss_seidel_ne", sp.csr_array(A.astype(np.float32)), 1, "forward", 1.0)
    H.save_spec(tmp_path / "a.npz", air)
    air2, _ = H.load_spec(tmp_path / "a.npz")
    assert np.array_equal(air2.levels[0].post.Fpts, L0.post.Fpts) and air2.levels[0].post.f_iterations == 2


def test_header_is_plain_c_and_links(tmp_path):
    """include/pyamg_amd.h compiles as C99 and a C translation unit that references every
    declared entry point links against libpyamg_amd.so (no call is made: no GPU here)."""
    import subprocess
    hdr = (ROOT / "include" / "pyamg_amd.h").read_text()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(pamg_[a-z0-9_]+)\s*\(", hdr_nc)))
    src = tmp_path / "abi.c"
    body = "\n".join(f"    tab[{i}] = (fn_t)&{n};" for i, n in enumerate(names))
    src.write_text(f'#include "pyamg_amd.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\nint main(void) {{\n'
                   f'    fn_t tab[{len(names)}];\n{body}\n'
                   f'    printf("%d %d\\n", {len(names)}, tab[0] != 0);\n    return 0;\n}}\n')
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{ROOT / 'include'}", str(src),
                        f"-L{capi.LIB_PATH.parent}", "-lpyamg_amd", f"-Wl,-rpath,{capi.LIB_PATH.parent}",
                        "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_batched_elasticity_assembly_matches_the_reference_gallery():
    """tools/problems.elasticity_p1_batched (the C5 workload generator) against the reference's element-by-element
    ``gallery.linear_elasticity_p1`` on a small tet mesh: same operator (1e-14 relative), same rigid-body modes, with and
    without clamped vertices"""
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not present")
    import pyamg
    import scipy.sparse as sp
    from tools.problems import cube_tet_mesh, elasticity_p1_batched
    V, T = cube_tet_mesh(6)
    A0, B0 = pyamg.gallery.linear_elasticity_p1(V, T, format="csr")
    A1, B1 = elasticity_p1_batched(V, T)
    assert A1.blocksize == (3, 3) and np.array_equal(B0, B1)
    assert abs(A0 - A1.tocsr()).max() <= 1e-14 * abs(A0).max()
    keep = np.flatnonzero(V[:, 0] > 0)
    k = np.flatnonzero(np.repeat(V[:, 0] > 0, 3))
    A2, B2 = elasticity_p1_batched(V, T, keep=keep)
    assert abs(sp.csr_array(A0[k][:, k]) - A2.tocsr()).max() <= 1e-14 * abs(A0).max() and np.array_equal(B0[k], B2)
    assert A2.indices.dtype == np.int32 and A2.indptr.dtype == np.int32


def test_pybind11_binding_module_surface():
    """The pybind11 face of Layer 1 (pyamg_amd/csrc/amg_core_bind.cpp, built by _build.py with plain g++): same
    function names as the ctypes twin, and -- like the reference's `.noconvert()` overload sets
    (relaxation_bind.cpp:708-715) -- a dtype mismatch is a TypeError, never a silent conversion.  No device needed."""
    import torch  # noqa: F401  (libamdhip64 binding order, as everywhere)
    from pyamg_amd import _build
    if not _build.pybind_path().exists():
        pytest.skip("pybind11 module not built")
    from pyamg_amd import _amg_core_pybind as pb
    from pyamg_amd import amg_core as ct
    assert set(ct.__all__) <= {n for n in dir(pb) if not n.startswith("_")}
    assert "pyamg_amd" in pb.version()
    Ap, Aj = np.zeros(2, dtype=np.int32), np.zeros(1, dtype=np.int32)
    with pytest.raises(TypeError):
        pb.gauss_seidel(Ap, Aj, np.zeros(1, dtype=np.float32), np.zeros(1), np.zeros(1), 0, 1, 1)          # mixed dtypes
    with pytest.raises(TypeError):
        pb.gauss_seidel(Ap.astype(np.int64), Aj, np.zeros(1), np.zeros(1), np.zeros(1), 0, 1, 1)           # int64 indices
    with pytest.raises(TypeError):
        pb.jacobi(Ap, Aj, np.zeros(1), np.zeros(1), np.zeros(1), np.zeros(1), 0, 1, 1, 1.0)                # omega is an array


def test_bench_headline_stays_small():
    """bench.py prints ONE compact line (the driver stopped parsing round 5's 20.8 KB line): built from a full record of a real run
    (profiles/r05_bench_n1.json) with every leg inflated, the headline must stay under bench.HEADLINE_LIMIT, be valid JSON of flat
    scalars inside `roofline` / `cpu_baseline` / `config`, and keep the contract's keys"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = json.loads((ROOT / "profiles" / "r05_bench_n1.json").read_text())
    # inflate: long strings, long arrays, more legs, more model rows
    out["residuals_gpu"] = [1.0] * 4000
    out["roofline"]["operator_stream"] = "x" * 5000
    out["roofline"]["kernel"] = "k" * 2000
    out["cpu_baseline"]["sample"] = "s" * 3000
    out["config"]["workload"] = "w" * 1000
    out["config"]["hierarchy_setup"] = "h" * 3000
    for i in range(12):
        out["extra"][f"leg{i}"] = dict(out["extra"]["c2"])
        out["extra"][f"leg{i}"]["hl_cg_s_to_tol"] = 0.1
    out["modelled_scaling"]["rows"] = out["modelled_scaling"]["rows"] * 3
    out["notes"] = ["n" * 500] * 10
    h = bench.headline(out)
    line = json.dumps(h)
    assert len(line) <= bench.HEADLINE_LIMIT < 10_000, len(line)
    h2 = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in h2, k
    for blk in ("roofline", "cpu_baseline", "config"):
        assert all(not isinstance(v, (dict, list)) for v in h2[blk].values()), blk
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in h2["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in h2["cpu_baseline"], k
    assert h2["value"] == out["value"] and h2["config"]["key"] == "c3"
    # the un-inflated record: everything fits, nothing is shed
    out2 = json.loads((ROOT / "profiles" / "r05_bench_n1.json").read_text())
    h3 = bench.headline(out2)
    assert len(json.dumps(h3)) <= 6_000 and "extra_c4_ms" in h3 and "modelled_ms_n8" in h3 and "extra_c5_block_gauss_seidel_ms" in h3
    # a failed multi-GPU run's line passes through
    hf = bench.headline({"metric": "m", "value": None, "error": "e" * 1000, "config": {"workload": "w"}})
    assert hf["value"] is None and len(hf["error"]) <= 300
