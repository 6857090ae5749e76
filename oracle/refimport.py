"""Put the prebuilt reference (oracle/_ref, see build_ref.py) on sys.path.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline + host setup).
``import oracle.refimport`` then ``import pyamg`` gives the reference package.
"""
import os
import sys
from pathlib import Path

REF_DIR = Path(__file__).resolve().parent / "_ref"

# the reference solve phase is serial; keep BLAS from threading the tiny dense coarse solve
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")


def available() -> bool:
    return (REF_DIR / "pyamg" / "__init__.pyc").exists()


def activate() -> bool:
    if not available():
        return False
    p = str(REF_DIR)
    if p not in sys.path:
        sys.path.insert(0, p)
    return True


activate()
