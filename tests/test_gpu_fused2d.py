"""Parity of the fused 2D Poisson HIP kernel against the CPU oracle -- runs on the MI355X box (-m gpu).

All calls go through the C ABI (hommx_amd.MicroCellPlan -> libhommx_hip.so).  Tolerance: the work is
float64; agreement with the oracle (sparse LU, energy form) is required to 1e-10 relative in the
Frobenius norm of each cell's tensor (observed ~1e-15..1e-13).
"""

import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(A, ref):
    return float(np.max(np.linalg.norm(A - ref, axis=(1, 2)) / np.linalg.norm(ref, axis=(1, 2))))


@pytest.fixture(scope="module")
def O():
    from oracle import hommx_oracle

    return hommx_oracle


def plan(dim, n, kind="poisson"):
    from hommx_amd import MicroCellPlan

    return MicroCellPlan(dim, n, kind)


@pytest.mark.parametrize("n", [3, 4, 5, 7, 8, 15, 16, 17, 24, 31, 32])
def test_random_coefficients_vs_oracle(n, rng, O):
    p = plan(2, n)
    assert p.kernel == "fused2d"
    nc = 5
    coef = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(nc, 2 * n * n)))
    M = np.eye(2)[None] + 0.4 * rng.standard_normal((nc, 2, 2))
    for MM in (None, M):
        A, info = p.solve(coef, MM, return_info=True)
        assert np.all(info == 0)
        assert relerr(A, O.effective_tensor_batch("poisson", 2, n, coef, MM)) < TOL


def test_golden_vectors():
    files = sorted(glob.glob(os.path.join(GOLDEN, "poisson2d_*.npz")))
    assert files
    for f in files:
        g = np.load(f)
        p = plan(int(g["dim"]), int(g["n"]))
        M = g["M"] if g["M"].size else None
        assert relerr(p.solve(g["coef"], M), g["A_eff"]) < TOL, f


def test_c1_laminate_closed_form():
    """C1 of BASELINE.json: every cell has the exact answer diag(harmonic, arithmetic mean)."""
    from hommx_amd import workloads as W

    msh, coef, _ = W.c1_laminate()
    A = plan(2, 16).solve(coef)
    assert np.abs(A - W.c1_exact(msh)).max() < 1e-12


def test_c3_full_size_closed_form_every_cell():
    """C3 at full size (128x128 macro = 32768 cells, 32x32 micro, stratified): closed form for EVERY cell."""
    from hommx_amd import workloads as W

    msh, coef, M = W.c3_wavy_laminate()
    assert coef.shape == (32768, 2048)
    A, info = plan(2, 32).solve(coef, M, return_info=True)
    assert np.all(info == 0)
    exact = W.stratified_laminate_exact(M)
    assert relerr(A, exact) < 1e-11


def test_c2_full_size_properties_and_sampled_oracle(O):
    """C2 at full size (8192 cells): sampled oracle parity + size-independent properties."""
    from hommx_amd import workloads as W

    msh, coef, _ = W.c2_inclusion()
    assert coef.shape == (8192, 2048)
    p = plan(2, 32)
    A = p.solve(coef)
    idx = np.linspace(0, 8191, 12).astype(int)
    assert relerr(A[idx], O.effective_tensor_batch("poisson", 2, 32, coef[idx])) < TOL
    # symmetry, Voigt-Reuss bounds (harmonic mean <= eig <= arithmetic mean)
    assert np.abs(A - np.transpose(A, (0, 2, 1))).max() < 1e-14
    ev = np.linalg.eigvalsh(A)
    assert np.all(ev[:, 0] >= 1.0 / np.mean(1.0 / coef, axis=1) * (1 - 1e-12))
    assert np.all(ev[:, 1] <= np.mean(coef, axis=1) * (1 + 1e-12))
    # linearity in the coefficient: A_H(s a) = s A_H(a)
    assert relerr(p.solve(3.0 * coef[:64]), 3.0 * A[:64]) < 1e-13
    # the disc is symmetric under y0 <-> y1 up to the mesh diagonal: A_H[0,0] == A_H[1,1]
    assert np.abs(A[:, 0, 0] - A[:, 1, 1]).max() < 1e-12
    # stratification with a rotation R: A_H -> R A_H R^T ... for M = R (orthogonal) Q = I, so A_H(M) = M A_H M^T
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    AR = p.solve(coef[:64], np.broadcast_to(R, (64, 2, 2)).copy())
    assert relerr(AR, R @ A[:64] @ R.T) < 1e-12


def test_constant_coefficient_and_high_contrast(rng, O):
    p = plan(2, 32)
    const = np.full((2, 2048), 0.7)
    assert np.abs(p.solve(const) - 0.7 * np.eye(2)).max() < 1e-14
    coef = np.where(rng.uniform(size=(4, 2048)) < 0.5, 1e-4, 1e3)
    A, info = p.solve(coef, return_info=True)
    assert np.all(info == 0)
    assert relerr(A, O.effective_tensor_batch("poisson", 2, 32, coef)) < 1e-8  # contrast 1e7


def test_edge_cases():
    from hommx_amd import _lib

    p = plan(2, 16)
    # empty batch
    A, info = p.solve(np.zeros((0, 512)), return_info=True)
    assert A.shape == (0, 2, 2) and info.shape == (0,)
    # ragged / wrong shapes are rejected on the host
    with pytest.raises(ValueError):
        p.solve(np.ones((3, 511)))
    with pytest.raises(ValueError):
        p.solve(np.ones((3, 512)), np.ones((2, 2, 2)))
    # numerical failure is reported per cell in info[], not raised (hmm.py:320-323): negative coefficient
    coef = np.ones((3, 512))
    coef[1] = -1.0
    A, info = p.solve(coef, return_info=True)
    assert info[0] == 0 and info[2] == 0 and info[1] > 0
    assert np.allclose(A[0], np.eye(2)) and np.allclose(A[2], np.eye(2))
    coef[2, 5] = np.nan
    A, info = p.solve(coef, return_info=True)
    assert info[2] > 0
    # maximum size of the fused family, and the first size outside it
    assert plan(2, 32).kernel == "fused2d"
    with pytest.raises(_lib.HommxLibraryError):
        plan(2, 2)


def test_device_pointer_entry_point_matches_host_entry_point(rng):
    import torch

    n, nc = 16, 33
    p = plan(2, n)
    coef = rng.uniform(0.1, 2.0, size=(nc, 2 * n * n))
    M = np.eye(2)[None] + 0.1 * rng.standard_normal((nc, 2, 2))
    ref = p.solve(coef, M)
    dc = torch.from_numpy(coef).cuda()
    dM = torch.from_numpy(M).cuda()
    out = torch.empty(nc, 2, 2, dtype=torch.float64, device="cuda")
    info = torch.zeros(nc, dtype=torch.int32, device="cuda")
    p.solve_device(nc, dc.data_ptr(), dM.data_ptr(), out.data_ptr(), info.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref)
