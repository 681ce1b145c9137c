"""LinearElasticityHMM / stratified / 3D PoissonHMM end to end on the GPU (-m gpu): macro solution vs the CPU-oracle twin."""

import numpy as np
import pytest

from hommx_amd import fem, hmm, mesh

pytestmark = pytest.mark.gpu


def _twin(h):
    from test_hmm_host import with_oracle

    return with_oracle(h)


def _rel(u, v):
    return np.linalg.norm(u.x.array - v.x.array) / np.linalg.norm(v.x.array)


def test_linear_elasticity_2d_beam():
    """test_integration_linear_elasticity.py:62-171 geometry (beam 1 x 0.2, 10x10 micro), mu = 5 + 4.5 sin 2 pi y0."""
    g = 0.4 * 0.2**2
    A = lambda x, y: hmm.Lame(1.25, 5.0 + 4.5 * np.sin(2 * np.pi * y[0]))

    def mk():
        msh = mesh.create_rectangle([(0, 0), (1.0, 0.2)], (20, 6))
        h = hmm.LinearElasticityHMM(msh, A, lambda x: np.array([0.0, -g]), mesh.create_unit_square(10, 10), 2.0**-6,
                                    petsc_options_cell_problem={"ksp_atol": 1e-9}, quadrature_degree=3)
        V = h.function_space
        clamp = fem.locate_dofs_topological(V, 1, fem.locate_entities_boundary(msh, 1, lambda x: np.isclose(x[0], 0)))
        h.set_boundary_conditions(fem.dirichletbc(np.zeros(2), clamp, V))
        return h

    h = mk()
    u = h.solve()
    assert h._plan.kernel == "small_wave" and np.all(h.cell_info == 0)
    assert _rel(u, _twin(mk()).solve()) < 1e-9
    assert u.x.array.reshape(-1, 2)[:, 1].min() < 0


def test_linear_elasticity_3d_constant_matches_plain_fem():
    """test_integration_linear_elasticity.py:205-322 with the tensor given as a full [3,3,3,3] array (Voigt kind)."""
    from test_hmm_host import _plain_elasticity_matrix

    msh = mesh.create_box([(0, 0, 0), (1.0, 0.2, 0.2)], (10, 3, 3))
    C = hmm.isotropic_hooke(1.25, 1.0, 3)
    g = 0.4 * 0.2**2
    h = hmm.LinearElasticityHMM(msh, lambda x, y: C, lambda x: np.array([0.0, 0.0, -g]), mesh.create_unit_cube(3, 3, 3), 1.0)
    V = h.function_space
    clamp = fem.locate_dofs_topological(V, 2, fem.locate_entities_boundary(msh, 2, lambda x: np.isclose(x[0], 0)))
    h.set_boundary_conditions(fem.dirichletbc(np.zeros(3), clamp, V))
    u = h.solve()
    assert h._plan.kind == "elasticity_voigt"
    K = _plain_elasticity_matrix(msh, 1.25, 1.0)
    assert np.linalg.norm((K - h._A).toarray()) / np.linalg.norm(K.toarray()) < 1e-4  # the reference's tolerance
    assert np.linalg.norm((K - h._A).toarray()) / np.linalg.norm(K.toarray()) < 1e-12  # what we actually get


def test_stratified_elasticity_3d_rotated_fibres_small():
    """rotated_fibers.py shape (LinearElasticityStratifiedHMM, 4^3 micro cells as in the example) with a full 3x3 Dtheta^T."""
    from hommx_amd import workloads as W

    Wd = 0.4

    def A(x, y):
        inside = W.wrapped_disc(y[1], y[2])
        return hmm.Lame(1.0, np.where(inside, 100.0, 0.001))

    def Dt(x):
        gam = 0.5 * np.pi * x[1] / Wd
        dg = 0.5 * np.pi / Wd
        Dth = np.array([[1.0, 0, 0], [0, 1.0, 0], [-np.sin(gam), dg * (-np.sin(gam) * x[2] - np.cos(gam) * x[0]), np.cos(gam)]])
        return Dth.T

    def mk():
        msh = mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], (5, 2, 1))
        h = hmm.LinearElasticityStratifiedHMM(msh, A, lambda x: np.array([0.0, 0.0, -0.05 * 0.4**2]),
                                              mesh.create_unit_cube(4, 4, 4), 2.0**-5, Dt)
        V = h.function_space
        clamp = fem.locate_dofs_topological(V, 2, fem.locate_entities_boundary(msh, 2, lambda x: np.isclose(x[0], 0)))
        h.set_boundary_conditions(fem.dirichletbc(np.zeros(3), clamp, V))
        return h

    h = mk()
    u = h.solve()
    assert np.all(h.cell_info == 0)
    assert _rel(u, _twin(mk()).solve()) < 1e-7


def test_poisson_3d():
    """test_integration_poisson.py:243-294 set-up (6^3 / 6^3, A = 1.1 + x0 + sin 2 pi y0, f = 1)."""
    A = lambda x, y: 1.1 + x[0] + np.sin(2 * np.pi * y[0])
    mk = lambda: hmm.PoissonHMM(mesh.create_unit_cube(6, 6, 6), A, lambda x: 1.0, mesh.create_unit_cube(6, 6, 6), 1 / 8,
                                petsc_options_cell_problem={"ksp_atol": 1e-9}, quadrature_degree=3)
    u = mk().solve()
    assert _rel(u, _twin(mk()).solve()) < 1e-9
    assert u.x.array.max() > 0 and np.isclose(u.x.array[0], 0.0)
