#!/usr/bin/env python3
"""Build the REAL reference (pyamg @ /root/reference) into ``oracle/_ref/``.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``pyamg_amd/``) may
import what this script produces; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg (and the host-side *setup* phase, which
the north-star leaves in the unmodified reference) use it.

What it does (recipe = SURVEY.md §8c, flags mirror the reference's
``meson.build:4,7``):

* compiles the reference's eight pybind11 extension modules
  ``pyamg/amg_core/*_bind.cpp`` **from the sources where they lie** under
  ``/root/reference`` with plain ``g++`` (no meson) into
  ``oracle/_ref/pyamg/amg_core/<name><EXT_SUFFIX>``;
* byte-compiles the reference's Python modules into *sourceless* ``.pyc`` files
  under ``oracle/_ref/pyamg/`` (build outputs only -- no reference source file
  is copied into this repository; ``oracle/_ref/`` is git-ignored but travels
  to the GPU box with the gpurun snapshot, like our own built ``.so``);
* writes the ``pyamg-<ver>.dist-info/METADATA`` stub that
  ``pyamg/__init__.py:12-13`` (``importlib.metadata.version``) needs.

Usage:  python oracle/build_ref.py [--force]
After it ran, ``import oracle.refimport`` puts ``oracle/_ref`` on ``sys.path``.
"""
from __future__ import annotations

import os
import py_compile
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get("PYAMG_REFERENCE", "/root/reference"))
DEST = HERE / "_ref"
VERSION = "5.3.0+reference"
MODULES = ["air", "evolution_strength", "graph", "krylov", "linalg",
           "relaxation", "ruge_stuben", "smoothed_aggregation"]
SKIP_DIRS = {"tests", "__pycache__"}
SKIP_FILES = {"bindthem.py"}


def _pybind_includes() -> list[str]:
    import pybind11
    return [f"-I{sysconfig.get_paths()['include']}", f"-I{pybind11.get_include()}"]


def _compile_ext(name: str, force: bool) -> str:
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    src = REF / "pyamg" / "amg_core" / f"{name}_bind.cpp"
    out = DEST / "pyamg" / "amg_core" / f"{name}{ext}"
    if out.exists() and not force and out.stat().st_mtime >= src.stat().st_mtime:
        return f"[ref] {name}: up to date"
    cmd = ["g++", "-O2", "-std=c++11", "-shared", "-fPIC", "-ftemplate-depth=2048",
           *_pybind_includes(), f"-I{REF / 'pyamg' / 'amg_core'}", str(src), "-o", str(out)]
    subprocess.run(cmd, check=True)
    return f"[ref] {name}: built"


def _bytecompile_tree() -> int:
    n = 0
    root = REF / "pyamg"
    for dirpath, dirnames, filenames in os.walk(root):
        dirnames[:] = [d for d in dirnames if d not in SKIP_DIRS]
        rel = Path(dirpath).relative_to(root)
        for fn in filenames:
            if not fn.endswith(".py") or fn in SKIP_FILES:
                continue
            out = DEST / "pyamg" / rel / (fn + "c")      # module.pyc beside nothing: sourceless import
            out.parent.mkdir(parents=True, exist_ok=True)
            py_compile.compile(str(Path(dirpath) / fn), cfile=str(out), doraise=True,
                               dfile=f"<reference>/pyamg/{rel / fn}")
            n += 1
    return n


def available() -> bool:
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    return all((DEST / "pyamg" / "amg_core" / f"{m}{ext}").exists() for m in MODULES) \
        and (DEST / "pyamg" / "__init__.pyc").exists()


def build(force: bool = False) -> bool:
    """Build oracle/_ref.  Returns False (and does nothing) when the reference
    tree is absent (e.g. on the GPU box, which only uses the prebuilt files)."""
    if not (REF / "pyamg" / "amg_core" / "relaxation_bind.cpp").exists():
        return available()
    (DEST / "pyamg" / "amg_core").mkdir(parents=True, exist_ok=True)
    n = _bytecompile_tree()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for msg in ex.map(lambda m: _compile_ext(m, force), MODULES):
            print(msg)
    di = DEST / f"pyamg-{VERSION}.dist-info"
    di.mkdir(exist_ok=True)
    (di / "METADATA").write_text(f"Metadata-Version: 2.1\nName: pyamg\nVersion: {VERSION}\n")
    (di / "INSTALLER").write_text("oracle/build_ref.py\n")
    print(f"[ref] byte-compiled {n} modules -> {DEST}")
    return available()


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    sys.exit(0 if ok else 1)
