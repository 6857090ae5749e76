"""Build recipe for libpyamg_amd.so (hand-written HIP for gfx950 only, built in-tree).

    python -m pyamg_amd._build [--force]

hipcc cross-compiles without a GPU.  ``-ffp-contract=off`` is part of the arithmetic
contract (no FMA contraction: results stay bit-identical to the reference's scalar loops).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libpyamg_amd.so"
SOURCES = ["pamg_matrix.hip", "pamg_solver.hip", "pamg_capi.hip", "pamg_setup.hip", "pamg_schwarz.hip", "pamg_dist.hip", "pamg_aggregate.hip", "pamg_lane.hip", "pamg_line.hip", "pamg_kz.hip", "pamg_blane.hip", "pamg_block.hip", "pamg_renumber.hip"]
HEADERS = ["pamg_common.h", "pamg_kernels.h", "pamg_tile_kernels.h", "pamg_tile_plan.h", "pamg_lane_plan.h", "pamg_lanem_plan.h", "pamg_schwarz_plan.h", "pamg_line_plan.h", "pamg_kz_plan.h", "pamg_blane_plan.h", "pamg_spg_plan.h", "pamg_stream_plan.h", "pamg_rowmask_map.h", "pamg_plan_vec.h", "pamg_host_threads.h", "amg_core_bind.cpp", "../../include/pyamg_amd.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libpyamg_amd.so)")


def stale() -> bool:
    if not LIB.exists():
        return True
    try:
        import pybind11  # noqa: F401
        if shutil.which("g++") and not pybind_path().exists():
            return True
    except Exception:       # pragma: no cover
        pass
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [(CSRC / h).resolve() for h in HEADERS] + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not stale():
        return LIB
    cc = hipcc()
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)

    def one(src: str) -> Path:
        obj = objdir / (src + ".o")
        cmd = [cc, *FLAGS, "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    build_pybind(verbose)
    return LIB


def pybind_path() -> Path:
    import sysconfig
    return PKG / ("_amg_core_pybind" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pybind(verbose: bool = True):
    """the pybind11 face of Layer 1 (csrc/amg_core_bind.cpp): plain g++ against libpyamg_amd.so.  Optional -- the
    ctypes twin (amg_core.py) covers the same surface; skipped when pybind11 or a host compiler is missing."""
    try:
        import pybind11
        import sysconfig
    except Exception:       # pragma: no cover
        return None
    cxx = shutil.which("g++") or shutil.which("c++")
    if not cxx:             # pragma: no cover
        return None
    out = pybind_path()
    src = CSRC / "amg_core_bind.cpp"
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
           str(src), "-o", str(out), f"-L{PKG}", "-l:libpyamg_amd.so", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
