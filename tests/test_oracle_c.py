"""The two CPU restatements pin each other: oracle/hommx_oracle_c.c (closed-form periodic stencil, dense block-cyclic elimination,
Schur form) against oracle/hommx_oracle.py (assembly from element gradients, sparse LU, energy form of hmm.py:652-667 / 774-789),
and both against the closed forms the reference's tests imply."""

import glob
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import hommx_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("n", [3, 4, 7, 16, 24, 32])
def test_c_oracle_matches_numpy_oracle(n, rng):
    coef = np.exp(rng.uniform(np.log(0.02), np.log(20.0), size=(3, 2 * n * n)))
    M = np.eye(2)[None] + 0.4 * rng.standard_normal((3, 2, 2))
    for MM in (None, M):
        got, info = c_oracle.effective_tensor_batch_c(n, coef, MM, threads=2, return_info=True)
        ref = O.effective_tensor_batch("poisson", 2, n, coef, MM)
        assert not info.any()
        assert np.max(np.linalg.norm(got - ref, axis=(1, 2)) / np.linalg.norm(ref, axis=(1, 2))) < 1e-12


def test_c_oracle_on_golden_vectors_and_closed_forms():
    for f in sorted(glob.glob(os.path.join(GOLDEN, "poisson2d_*.npz"))):
        g = np.load(f)
        M = g["M"] if g["M"].size else None
        got = c_oracle.effective_tensor_batch_c(int(g["n"]), g["coef"], M)
        assert np.abs(got - g["A_eff"]).max() < 1e-12 * np.abs(g["A_eff"]).max(), f
    # laminate a in {5, 0.05} switching at y0 = 1/4, 3/4 (laminate.py:101-102): A_H = diag(harmonic, arithmetic mean), exactly
    from hommx_amd import workloads as W

    n = 16
    y = W.element_barycentres(2, n)
    coef = np.where(np.cos(2 * np.pi * y[:, 0]) < 0, 5.0, 0.05)[None]
    A = c_oracle.effective_tensor_batch_c(n, coef)[0]
    assert abs(A[0, 0] - 2.0 / (1 / 5.0 + 1 / 0.05)) < 1e-13 and abs(A[1, 1] - 2.525) < 1e-13 and abs(A[0, 1]) < 1e-13
    # a non-positive coefficient is reported, not solved
    bad = np.ones((2, 2 * n * n))
    bad[1] = -1.0
    A, info = c_oracle.effective_tensor_batch_c(n, bad, return_info=True)
    assert info[0] == 0 and info[1] > 0 and np.isnan(A[1]).all()
    with pytest.raises(ValueError):
        c_oracle.effective_tensor_batch_c(n, bad[:, :-1])


@pytest.mark.parametrize("kind,dim,n,layout", [("poisson", 3, 4, "scalar"), ("poisson", 3, 5, "matrix"), ("poisson", 2, 7, "matrix"),
                                               ("elasticity", 2, 6, "lame"), ("elasticity", 2, 5, "hooke"), ("elasticity", 3, 3, "lame"),
                                               ("elasticity", 3, 4, "lame"), ("elasticity", 3, 3, "hooke")])
def test_generic_c_restatement_matches_numpy_oracle(kind, dim, n, layout, rng):
    """Element-by-element C restatement (hommx_oracle_generic: loops over tensor indices, dense Cholesky) against the NumPy oracle (einsum
    assembly, sparse LU, energy form) for the 3D and elasticity paths -- hmm.py:644-667 / 759-789 / 887-922 / 1024-1067, with and without
    the stratification matrix M (e_D(u) = sym(M . nabla_grad u), hmm.py:1024-1030)."""
    n_el = (2 if dim == 2 else 6) * n**dim
    if layout == "scalar":
        coef = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=n_el))
    elif layout == "matrix":
        G = rng.uniform(-1, 1, size=(n_el, dim, dim))
        coef = 0.3 * G @ np.swapaxes(G, -1, -2) + 0.5 * np.eye(dim)
    elif layout == "lame":
        coef = rng.uniform(0.3, 3.0, size=(n_el, 2))
    else:  # general Hooke tensor with the symmetries of elasticity, positive definite on symmetric strains
        coef = O.isotropic_hooke(rng.uniform(0.5, 2.0, size=n_el), rng.uniform(0.5, 2.0, size=n_el), dim)
        S = rng.uniform(-0.2, 0.2, size=(n_el, dim, dim))
        S = 0.5 * (S + np.swapaxes(S, -1, -2))
        coef = coef + np.einsum("eij,ekl->eijkl", S, S)
    M = np.eye(dim) + 0.3 * rng.standard_normal((dim, dim))
    for MM in (None, M):
        ref = O.effective_tensor(O.build_cell_problem(kind, dim, n, coef, MM))
        got = c_oracle.effective_tensor_generic_c(kind, dim, n, coef, MM)
        assert np.abs(got - ref).max() < 1e-11 * np.abs(ref).max(), (kind, dim, n, layout, np.abs(got - ref).max() / np.abs(ref).max())
    with pytest.raises(ValueError):
        c_oracle.effective_tensor_generic_c(kind, dim, n, -np.abs(coef))


def test_generic_c_restatement_on_known_answers():
    """Layered elastic medium, lambda = 1, mu in {5, 0.5} switching at y0 = 1/4, 3/4: C_H[00,00] = 1 / <1 / (lambda + 2 mu)>,
    shear E01 : C_H : E01 = 1 / <1 / mu> -- exact for P1 on aligned layers; and a constant Hooke tensor comes back unchanged
    (test_integration_linear_elasticity.py:205-322)."""
    from hommx_amd import workloads as W

    n = 4
    y = W.element_barycentres(3, n)
    mu = np.where((y[:, 0] > 0.25) & (y[:, 0] < 0.75), 5.0, 0.5)
    C = c_oracle.effective_tensor_generic_c("elasticity", 3, n, np.stack([np.ones_like(mu), mu], axis=1))
    assert abs(C[0, 0] - 1.0 / np.mean(1.0 / (1.0 + 2.0 * np.array([5.0, 0.5])))) < 1e-12
    assert abs(C[3, 3] - 1.0 / np.mean(1.0 / np.array([5.0, 0.5]))) < 1e-12      # Voigt index 3 = (0, 1), tensorial unit strain
    const = c_oracle.effective_tensor_generic_c("elasticity", 3, 3, np.tile([1.25, 1.0], (6 * 27, 1)))
    ref = O.effective_tensor(O.build_cell_problem("elasticity", 3, 3, np.tile([1.25, 1.0], (6 * 27, 1))))
    assert np.abs(const - ref).max() < 1e-13
    assert abs(const[0, 0] - 3.25) < 1e-13 and abs(const[0, 1] - 1.25) < 1e-13 and abs(const[3, 3] - 1.0) < 1e-13
