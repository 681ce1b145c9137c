"""The reference's integration tests, re-expressed on hommx_amd's solver classes with the REAL GPU path (-m gpu).

Final-macro-solution parity: the same class with the CPU oracle injected for the micro solves must give the same
macro solution (rel. L2 <= 1e-9).
"""

import numpy as np
import pytest

from hommx_amd import fem, hmm, mesh

pytestmark = pytest.mark.gpu


def _oracle_twin(h):
    from test_hmm_host import with_oracle

    return with_oracle(h)


def _rel_l2(u, v):
    d = u.copy()
    d.x.array[:] -= v.x.array
    zero = lambda x: 0 * x[0]
    return np.sqrt(fem.l2_error_squared(d, zero) / fem.l2_error_squared(v, zero))


def test_analytical_example_1_gpu():
    """test_integration_poisson.py:121-143."""
    A = lambda x, y: 1.0 / (2.0 + np.cos(2 * np.pi * y[0]))
    f = lambda x: np.pi**2 * (0.5 + 1 / np.sqrt(3)) * np.sin(np.pi * x[0]) * np.sin(np.pi * x[1])
    mk = lambda: hmm.PoissonHMM(mesh.create_unit_square(15, 15), A, f, mesh.create_unit_square(15, 15), 0.1 / 15,
                                petsc_options_cell_problem={"ksp_atol": 1e-10}, quadrature_degree=3)
    h = mk()
    u = h.solve()
    assert h._plan.kernel == "fused2d" and np.all(h.cell_info == 0)
    assert np.isclose(fem.l2_error_squared(u, lambda x: np.sin(np.pi * x[0]) * np.sin(np.pi * x[1])), 0, atol=5e-5)
    assert np.abs(h.effective_tensors[:, 1, 1] - 1 / np.sqrt(3)).max() < 1e-9
    u_cpu = _oracle_twin(mk()).solve()
    assert _rel_l2(u, u_cpu) < 1e-9


def test_periodic_poisson_hmm_matches_periodic_homogenization_gpu():
    """test_integration_poisson.py:188-240."""
    A_y = lambda y: 2.0 + np.sin(2 * np.pi * y[0])
    msh, mic = mesh.create_unit_square(15, 15), mesh.create_unit_square(15, 15)
    h = hmm.PoissonHMM(msh, lambda x, y: A_y(y), lambda x: 1.0, mic, 0.1 / 15, quadrature_degree=3)
    u_hmm = h.solve()
    per = hmm.PoissonPeriodicHMM(msh, A_y, lambda x: 1.0, mic, 0.1 / 15, quadrature_degree=3)
    per.set_boundary_conditions(h._bcs)
    u_per = per.solve()
    assert np.linalg.norm((h._A - per._lp_A).toarray()) < 1e-8
    d = u_hmm.copy()
    d.x.array[:] -= u_per.x.array
    assert np.sqrt(fem.l2_error_squared(d, lambda x: 0 * x[0])) < 1e-12


def test_stratified_darcy_flow_gpu():
    """PoissonStratifiedHMM with a square 2x2 Dtheta^T (test_integration_poisson.py:481-572 convention),
    Darcy boundary data (inclusion.py:64-88): GPU solution == CPU-oracle solution."""
    A = lambda x, y: np.where(np.cos(2 * np.pi * y[1]) < 0, 5.0, 0.05)
    Dt = lambda x: np.array([[1.0, -2 * np.pi * np.cos(2 * np.pi * x[0])], [0.0, 1.0]])

    def mk():
        msh = mesh.create_unit_square(12, 12)
        h = hmm.PoissonStratifiedHMM(msh, A, lambda x: 1.0, mesh.create_unit_square(16, 16), 2.0**-5, Dt)
        V = h.function_space
        left = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0.0))
        right = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 1.0))
        h.set_boundary_conditions([fem.dirichletbc(1.0, left, V), fem.dirichletbc(0.0, right, V)])
        return h

    h = mk()
    u = h.solve()
    from hommx_amd import workloads as W

    M = np.stack([Dt(c) for c in h._msh.cell_midpoints()])
    assert np.abs(h.effective_tensors - W.stratified_laminate_exact(M)).max() < 1e-11
    assert _rel_l2(u, _oracle_twin(mk()).solve()) < 1e-9
    assert u.x.array.min() >= -1e-9  # maximum principle (f >= 0, boundary data in [0, 1])


def test_single_cell_seam_matches_batch():
    """_compute_local_stiffness(cell) (the reference seam, hmm.py:334) == the batched assembly."""
    A = lambda x, y: 1.0 + x[0] + 0.5 * np.sin(2 * np.pi * y[1])
    h = hmm.PoissonHMM(mesh.create_unit_square(3, 3), A, lambda x: 1.0, mesh.create_unit_square(8, 8), 0.01,
                       quadrature_degree=3)
    h.solve()
    cells = np.arange(h._msh.num_cells)
    S = h._local_stiffness_from_tensors(cells, h.effective_tensors)
    for c in (0, 7, 17):
        assert np.allclose(h._compute_local_stiffness(c), S[c], rtol=1e-13, atol=0)
