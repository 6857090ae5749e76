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


def test_extract_reads_back_smoother_parameters():
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
    ml3 = pyamg.smoothed_aggregation_solver(A, max_coarse=10, presmoother="schwarz", postsmoother="schwarz")
    with pytest.raises(NotImplementedError):
        H.extract(ml3)
    ml4 = pyamg.smoothed_aggregation_solver(A, max_coarse=10, coarse_solver="cg")
    with pytest.raises(NotImplementedError):
        H.extract(ml4)


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
