"""Corrector output path: chi_m from the GPU back substitution == oracle correctors (-m gpu)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _center(chi, bs):
    x = chi.reshape(chi.shape[0], -1, bs)
    return (x - x.mean(axis=1, keepdims=True)).reshape(chi.shape)


@pytest.mark.parametrize("kind,dim,n", [("poisson", 2, 8), ("poisson", 2, 32), ("elasticity", 2, 6), ("poisson", 3, 4), ("elasticity", 3, 3)])
def test_correctors_vs_oracle(kind, dim, n, rng):
    from hommx_amd import MicroCellPlan
    from oracle import hommx_oracle as O

    n_el = (2 if dim == 2 else 6) * n**dim
    nc = 3
    if kind == "poisson":
        coef = np.exp(rng.uniform(np.log(0.1), np.log(5.0), size=(nc, n_el)))
    else:
        coef = np.stack([rng.uniform(0.5, 2.0, (nc, n_el)), np.exp(rng.uniform(np.log(0.1), np.log(10.0), (nc, n_el)))], axis=-1)
    M = np.eye(dim)[None] + 0.3 * rng.standard_normal((nc, dim, dim))
    bs = 1 if kind == "poisson" else dim
    p = MicroCellPlan(dim, n, kind)
    for MM in (None, M):
        A, corr, info = p.solve(coef, MM, return_info=True, return_correctors=True)
        assert np.all(info == 0)
        A2 = p.solve(coef, MM)
        assert np.abs(A - A2).max() < 1e-10 * np.abs(A2).max()
        for c in range(nc):
            cp = O.build_cell_problem(kind, dim, n, coef[c], None if MM is None else MM[c])
            chi = _center(O.solve_correctors(cp).T, bs)  # [t, n_dof]
            assert np.abs(corr[c] - chi).max() < 1e-9 * max(1e-30, np.abs(chi).max())
            # the correctors reproduce the effective tensor through the energy functional of the reference
            AH = O.effective_tensor(cp, corr[c].T, form="energy")
            assert np.abs(AH - A[c]).max() < 1e-10 * np.abs(A[c]).max()


def test_periodic_hmm_correctors_and_macro_basis_correctors():
    """PoissonPeriodicHMM.correctors (hmm.py:1211-1245) and the per-basis-function correctors of BaseHMM (hmm.py:354-358)."""
    from hommx_amd import hmm, mesh
    from oracle import hommx_oracle as O

    A_y = lambda y: 2.0 + np.sin(2 * np.pi * y[0]) * np.cos(2 * np.pi * y[1])
    msh, mic = mesh.create_unit_square(3, 3), mesh.create_unit_square(12, 12)
    per = hmm.PoissonPeriodicHMM(msh, A_y, lambda x: 1.0, mic, 0.05, quadrature_degree=3)
    AH = per.compute_effective_tensor()
    coef = O.sample_coefficient(lambda x, y: A_y(y), np.zeros(2), 2, 12, 3)
    cp = O.build_cell_problem("poisson", 2, 12, coef)
    assert np.abs(AH - O.effective_tensor(cp)).max() < 1e-12
    chi = O.solve_correctors(cp)
    pm = O.periodic_master_map(2, 12)
    for q, f in enumerate(per.correctors):
        ref = chi[:, q] - chi[:, q].mean()
        assert np.abs(f.x.array - ref[pm]).max() < 1e-10
        # periodic: opposite faces agree
        v = f.x.array.reshape(13, 13)
        assert np.abs(v[0] - v[-1]).max() == 0 and np.abs(v[:, 0] - v[:, -1]).max() == 0
    # macro basis function correctors: corrector_i = eps * grad(phi_i) . chi
    h = hmm.PoissonHMM(msh, lambda x, y: A_y(y), lambda x: 1.0, mic, 0.05, quadrature_degree=3)
    cs = h.correctors_for_cell(4)
    assert len(cs) == 3
    X = msh.cell_vertices()[4]
    G = O.p1_gradients(X)
    for i, f in enumerate(cs):
        ref = 0.05 * (chi - chi.mean(axis=0)) @ G[i]
        assert np.abs(f.x.array - ref[pm]).max() < 1e-10


def test_periodic_linear_problem_front_end(rng):
    """hommx_amd.cell_problem.PeriodicLinearProblem: one cell problem, all canonical loads, functions on the micro mesh."""
    from hommx_amd import fem, mesh
    from hommx_amd.cell_problem import PeriodicLinearProblem, create_periodic_boundary_conditions
    from oracle import hommx_oracle as O

    n = 9
    mic = mesh.create_unit_square(n, n)
    V = fem.functionspace(mic, ("Lagrange", 1))
    mpc = create_periodic_boundary_conditions(V)
    coef = rng.uniform(0.2, 3.0, 2 * n * n)
    prob = PeriodicLinearProblem("poisson", coef, mpc)
    sols = prob.solve()
    cp = O.build_cell_problem("poisson", 2, n, coef)
    assert prob.info == 0 and np.abs(prob.effective_tensor - O.effective_tensor(cp)).max() < 1e-12
    chi = O.solve_correctors(cp)
    for m, f in enumerate(sols):
        ref = (chi[:, m] - chi[:, m].mean())[mpc.to_periodic]
        assert np.abs(f.x.array - ref).max() < 1e-10
        assert np.array_equal(f.x.array[mpc.slaves], f.x.array[mpc.masters])  # u(x, 1) = u(x, 0) etc.
