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
