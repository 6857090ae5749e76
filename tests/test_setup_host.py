"""CPU tests of the setup-phase host logic (pyamg_amd/aggregation.py): the restart loop of approximate_spectral_radius
with a NumPy stand-in that has the semantics of the device Arnoldi (pamg_arnoldi_*: all steps are run, the process is
truncated at the first breakdown, the restart vector stays with the process), checked against the reference
(oracle/_ref); and the patching of a reference package by device_setup().  No device work here."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

import pyamg_amd.aggregation as ag


class NumpyArnoldi:
    def __init__(self, A, maxiter):
        self.A, self.n = A, A.shape[0]
        self.m = min(self.n, maxiter)
        self.v0, self.planes = None, 0

    def run(self, v0, breakdown):
        if v0 is not None:
            self.v0 = np.ravel(v0).copy()
            self.planes = 2 if np.iscomplexobj(v0) else 1
        V = [self.v0 / np.sqrt(np.vdot(self.v0, self.v0).real)]
        m = self.m
        H = np.zeros((m + 1, m), dtype=complex if self.planes == 2 else float)
        with np.errstate(all="ignore"):
            for j in range(m):
                w = self.A @ V[-1]
                for i, u in enumerate(V):
                    H[i, j] = np.vdot(u, w)
                    w = w - H[i, j] * u
                H[j + 1, j] = np.sqrt(np.vdot(w, w).real)
                V.append(w / H[j + 1, j])
        nc, flag = m, False
        for j in range(m):
            if not (H[j + 1, j].real >= breakdown):
                nc, flag = j + 1, True
                break
        self.V = V
        return H, nc, flag

    def combine(self, coef):
        coef = np.ravel(coef)
        self.v0 = sum(c * v for c, v in zip(coef, self.V[:len(coef)]))
        if np.iscomplexobj(coef):
            self.planes = 2

    def vector(self):
        return self.v0.reshape(-1, 1)

    def free(self):
        pass


def _reference():
    import oracle.refimport as ri
    if not ri.available():
        pytest.skip("oracle/_ref not built")
    import pyamg
    return pyamg


def test_restart_loop_against_the_reference(monkeypatch):
    pyamg = _reference()
    from pyamg.util.linalg import approximate_spectral_radius as ref_rho
    monkeypatch.setattr(ag, "_Arnoldi", NumpyArnoldi)
    rng = np.random.default_rng(3)
    n = 400
    K = sp.diags_array([np.ones(n - 1), -np.ones(n - 1)], offsets=[1, -1], format="csr") * 3.0
    cases = [pyamg.gallery.poisson((30, 30), format="csr"),
             (K + sp.random_array((n, n), density=0.01, random_state=rng, format="csr") * 0.1 + 0.05 * sp.eye_array(n)).tocsr(),   # complex Ritz pair
             sp.csr_array(np.diag([1.0, 2.0, 3.0])),             # maxiter clipped to n, breakdown
             sp.csr_array(np.eye(2))]
    for k, M in enumerate(cases):
        for kw in ({}, {"maxiter": 8, "restart": 2}, {"tol": 1e-6, "maxiter": 20, "restart": 8}):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                np.random.seed(17 + k)
                r_ref = ref_rho(M.copy(), **kw)
                np.random.seed(17 + k)
                v0 = np.random.rand(M.shape[0], 1)
                r = ag._spectral_radius(M, kw.get("tol", 0.01), kw.get("maxiter", 15), kw.get("restart", 5), v0)
            assert abs(r - r_ref) <= 1e-12 * abs(r_ref), (k, kw, r, r_ref)


def test_device_setup_patches_and_restores():
    pyamg = _reference()
    import pyamg.aggregation.aggregation as agg
    import pyamg.relaxation.smoothing as smoothing
    before = (agg.jacobi_prolongation_smoother, agg.richardson_prolongation_smoother, smoothing.approximate_spectral_radius)
    with ag.device_setup(pyamg):
        assert agg.jacobi_prolongation_smoother is ag.jacobi_prolongation_smoother
        assert agg.richardson_prolongation_smoother is ag.richardson_prolongation_smoother
        assert smoothing.approximate_spectral_radius is not before[2]
        # operands the device path does not take stay with the reference function (a LinearOperator's matvec is host code)
        from scipy.sparse.linalg import aslinearoperator
        np.random.seed(0)
        rho = smoothing.approximate_spectral_radius(aslinearoperator(sp.csr_array(np.diag([1.0, 2.0, 3.0]))))
        assert abs(rho - 3.0) < 1e-12
    assert before == (agg.jacobi_prolongation_smoother, agg.richardson_prolongation_smoother, smoothing.approximate_spectral_radius)
    with pytest.raises(RuntimeError):
        with ag.device_setup(pyamg):
            raise RuntimeError("x")
    assert agg.jacobi_prolongation_smoother is before[0]


def test_argument_checks_need_no_device():
    A = sp.csr_array(np.eye(3))
    with pytest.raises(ValueError):
        ag.approximate_spectral_radius(sp.csr_array((3, 4)))
    with pytest.raises(ValueError):
        ag.approximate_spectral_radius(A, maxiter=0)
    with pytest.raises(ValueError):
        ag.approximate_spectral_radius(A, restart=-1)
    with pytest.raises(NotImplementedError):
        ag.approximate_spectral_radius(A.astype(np.complex128))
    A.rho = 42.0
    assert ag.approximate_spectral_radius(A) == 42.0                 # cached value wins, like the reference
    with pytest.raises(NotImplementedError):
        ag.jacobi_prolongation_smoother(sp.csr_array(np.eye(3)), sp.csr_array(np.eye(3)), None, None, filter_entries=True)
    with pytest.raises(ValueError):
        ag.jacobi_prolongation_smoother(sp.csr_array(np.eye(3)), sp.csr_array(np.eye(3)), None, None, weighting="nope")
