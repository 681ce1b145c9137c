"""On-device sampling of two-phase coefficients == the element-stream path, bit for bit (-m gpu)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fused_two_phase_equals_stream_c2():
    from hommx_amd import MicroCellPlan, workloads as W

    msh, coef, _ = W.c2_inclusion(nx=16, n=32)
    y = W.element_barycentres(2, 32)
    mask = W.wrapped_disc(y[:, 0], y[:, 1])
    c = msh.cell_midpoints()
    values = np.stack([np.full(len(c), 0.1), 0.001 * (1 + 9 * c[:, 0])], axis=1)
    p = MicroCellPlan(2, 32, "poisson")
    A, info = p.solve_two_phase(mask, values, return_info=True)
    assert np.all(info == 0)
    assert np.array_equal(A, p.solve(coef))


@pytest.mark.parametrize("kind,dim,n", [("poisson", 2, 15), ("poisson", 3, 5), ("elasticity", 2, 6), ("elasticity", 3, 4)])
def test_two_phase_equals_stream(kind, dim, n, rng):
    from hommx_amd import MicroCellPlan

    p = MicroCellPlan(dim, n, kind)
    nc = 5
    mask = rng.uniform(size=p.n_el) < 0.4
    ncomp = p.n_comp
    values = rng.uniform(0.2, 4.0, size=(nc, 2, ncomp)) if ncomp > 1 else rng.uniform(0.2, 4.0, size=(nc, 2))
    M = np.eye(dim)[None] + 0.3 * rng.standard_normal((nc, dim, dim))
    coef = np.where(mask[None, :, None], values[:, 1:2].reshape(nc, 1, -1), values[:, 0:1].reshape(nc, 1, -1))
    coef = coef.reshape((nc, p.n_el) + ((ncomp,) if ncomp > 1 else ()))
    for MM in (None, M):
        assert np.array_equal(p.solve_two_phase(mask, values, MM), p.solve(coef, MM))
    with pytest.raises(ValueError):
        p.solve_two_phase(mask[:-1], values)


def test_two_phase_coefficient_object_in_solver_classes():
    """hmm.TwoPhase through PoissonStratifiedHMM and LinearElasticityHMM == the generic callable path."""
    from hommx_amd import fem, hmm, mesh, workloads as W

    ind = lambda y: np.cos(2 * np.pi * y[1]) < 0
    Dt = lambda x: np.array([[1.0, -2 * np.pi * np.cos(2 * np.pi * x[0])], [0.0, 1.0]])
    tp = hmm.TwoPhase(ind, lambda x: 5.0 * (1 + x[0]), lambda x: 0.05)
    gen = lambda x, y: np.where(ind(y), 5.0 * (1 + x[0]), 0.05)
    us = []
    for A in (tp, gen):
        h = hmm.PoissonStratifiedHMM(mesh.create_unit_square(8, 8), A, lambda x: 1.0, mesh.create_unit_square(16, 16), 2.0**-5, Dt)
        us.append(h.solve().x.array.copy())
    assert np.array_equal(us[0], us[1])
    fib = lambda y: W.wrapped_disc(y[1], y[2])
    tp3 = hmm.TwoPhase(fib, lambda x: hmm.Lame(1.0, 100.0 * (1 + x[0])), lambda x: hmm.Lame(1.0, 0.001))
    gen3 = lambda x, y: hmm.Lame(1.0, np.where(fib(y), 100.0 * (1 + x[0]), 0.001))
    us = []
    for A in (tp3, gen3):
        msh = mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], (3, 1, 1))
        h = hmm.LinearElasticityHMM(msh, A, lambda x: np.array([0, 0, -0.008]), mesh.create_unit_cube(4, 4, 4), 0.03)
        V = h.function_space
        clamp = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0))
        h.set_boundary_conditions(fem.dirichletbc(np.zeros(3), clamp, V))
        us.append(h.solve().x.array.copy())
    assert np.array_equal(us[0], us[1])
