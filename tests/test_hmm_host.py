"""Host logic of the solver classes (macro assembly, BC lifting, solve) on the CPU.

The GPU plan is replaced by a stub that answers with the CPU oracle, so these tests exercise everything
ABOVE the C ABI and read like the reference's own integration tests (test/integration/*.py).
"""

import numpy as np
import pytest

from hommx_amd import fem, hmm, mesh
from oracle import hommx_oracle as O


class OraclePlan:
    def __init__(self, dim, n, kind):
        self.dim, self.n, self.kind = dim, n, kind
        self.t = dim if kind.startswith("poisson") else dim * (dim + 1) // 2

    def solve(self, coef, M=None, return_info=False, return_correctors=False):
        okind = "poisson" if self.kind.startswith("poisson") else "elasticity"
        c = coef
        if self.kind == "poisson_matrix":
            d = self.dim
            pairs = hmm._VOIGT[d]
            full = np.zeros(coef.shape[:2] + (d, d))
            for m, (i, j) in enumerate(pairs):
                full[..., i, j] = full[..., j, i] = coef[..., m]
            c = full
        if self.kind == "elasticity_voigt":
            raise NotImplementedError
        out = O.effective_tensor_batch(okind, self.dim, self.n, c, M)
        info = np.zeros(len(coef), np.int32)
        if return_correctors:
            bs = 1 if okind == "poisson" else self.dim
            corr = []
            for k in range(len(coef)):
                chi = O.solve_correctors(O.build_cell_problem(okind, self.dim, self.n, c[k], None if M is None else M[k])).T
                x = chi.reshape(chi.shape[0], -1, bs)
                corr.append((x - x.mean(axis=1, keepdims=True)).reshape(chi.shape))
            corr = np.stack(corr)
            return (out, corr, info) if return_info else (out, corr)
        return (out, info) if return_info else out


def with_oracle(h):
    kind = {"poisson": "poisson", "elasticity": "elasticity"}[h._kind]
    h._plan = OraclePlan(h._tdim, h._n_micro, kind)
    return h


def test_element_means_match_oracle_sampling():
    A = lambda x, y: 0.33 + 0.15 * (np.sin(2 * np.pi * x[0]) + np.sin(2 * np.pi * y[0]))
    msh = mesh.create_unit_square(2, 2)
    h = hmm.PoissonHMM(msh, A, lambda x: 1.0, mesh.create_unit_square(6, 6), 0.01, quadrature_degree=3)
    coef, kind = h._element_means(np.arange(msh.num_cells))
    assert kind == "poisson"
    for c in range(msh.num_cells):
        assert np.allclose(coef[c], O.sample_coefficient(A, msh.cell_midpoints()[c], 2, 6, 3), rtol=0, atol=1e-15)


def test_analytical_example_1():
    """test_integration_poisson.py:121-143 (15x15 / 15x15, squared L2 error < 5e-5)."""
    A = lambda x, y: 1.0 / (2.0 + np.cos(2 * np.pi * y[0]))
    f = lambda x: np.pi**2 * (0.5 + 1 / np.sqrt(3)) * np.sin(np.pi * x[0]) * np.sin(np.pi * x[1])
    msh, mic = mesh.create_unit_square(15, 15), mesh.create_unit_square(15, 15)
    h = with_oracle(hmm.PoissonHMM(msh, A, f, mic, 0.1 / 15, petsc_options_cell_problem={"ksp_atol": 1e-10},
                                   quadrature_degree=3))
    u = h.solve()
    err = fem.l2_error_squared(u, lambda x: np.sin(np.pi * x[0]) * np.sin(np.pi * x[1]))
    assert np.isclose(err, 0, atol=5e-5), err


def test_analytical_example_2():
    """test_integration_poisson.py:146-185."""
    A = lambda x, y: 0.33 + 0.15 * (np.sin(2 * np.pi * x[0]) + np.sin(2 * np.pi * y[0]))
    s, c, pi = np.sin, np.cos, np.pi

    def f(x):
        q = (0.454545454545455 * s(2 * pi * x[0]) + 1) ** 2 - 0.206611570247934
        return (3.25696945235949 * np.sqrt(q) * s(pi * x[0]) * s(pi * x[1])
                + pi**2 * (0.15 * s(2 * pi * x[0]) + 0.33) * s(pi * x[0]) * s(pi * x[1])
                - 2.96088132032681 * (0.454545454545455 * s(2 * pi * x[0]) + 1) * s(pi * x[1]) * c(pi * x[0])
                * c(2 * pi * x[0]) / np.sqrt(q))

    msh, mic = mesh.create_unit_square(15, 15), mesh.create_unit_square(15, 15)
    h = with_oracle(hmm.PoissonHMM(msh, A, f, mic, 0.1 / 15, quadrature_degree=3))
    u = h.solve()
    err = fem.l2_error_squared(u, lambda x: s(pi * x[0]) * s(pi * x[1]))
    assert np.isclose(err, 0, atol=5e-5), err


def test_periodic_poisson_hmm_matches_periodic_homogenization():
    """test_integration_poisson.py:188-240: HMM matrix == FEM matrix with constant A_hom (||.||_F < 1e-8, ||du|| < 1e-12)."""
    A_y = lambda y: 2.0 + np.sin(2 * np.pi * y[0])
    msh, mic = mesh.create_unit_square(15, 15), mesh.create_unit_square(15, 15)
    h = with_oracle(hmm.PoissonHMM(msh, lambda x, y: A_y(y), lambda x: 1.0, mic, 0.1 / 15, quadrature_degree=3))
    bnd = fem.locate_dofs_geometrical(h.function_space, lambda x: np.isclose(x[0], 0) | np.isclose(x[0], 1)
                                      | np.isclose(x[1], 0) | np.isclose(x[1], 1))
    h.set_boundary_conditions([fem.dirichletbc(0.0, bnd, h.function_space)])
    u_hmm = h.solve()
    per = hmm.PoissonPeriodicHMM(msh, A_y, lambda x: 1.0, mic, 0.1 / 15, quadrature_degree=3)
    with_oracle(per._inner)
    per.set_boundary_conditions([fem.dirichletbc(0.0, bnd, per.function_space)])
    u_per = per.solve()
    assert np.linalg.norm((h._A - per._lp_A).toarray()) < 1e-8
    d = u_hmm.copy()
    d.x.array[:] -= u_per.x.array
    assert np.sqrt(fem.l2_error_squared(d, lambda x: 0 * x[0])) < 1e-12


def _plain_elasticity_matrix(msh, lam, mu):
    d = msh.topology.dim
    X = msh.cell_vertices()
    ones = np.ones(X.shape[:2] + (1,))
    G = np.transpose(np.linalg.inv(np.concatenate([ones, X], axis=2))[:, 1:, :], (0, 2, 1))
    vol = msh.cell_volumes()
    C = O.isotropic_hooke(lam, mu, d)
    I = np.eye(d)
    eps = 0.5 * (np.einsum("pi,caj->capij", I, G) + np.einsum("pj,cai->capij", I, G))
    Ke = np.einsum("c,capij,ijkl,cbqkl->capbq", vol, eps, C, eps).reshape(len(vol), (d + 1) * d, (d + 1) * d)
    dofs = hmm._unroll_dofs(msh.cells.astype(np.int64), d)
    nb = dofs.shape[1]
    import scipy.sparse as sp

    N = msh.num_vertices * d
    return sp.coo_matrix((Ke.ravel(), (np.repeat(dofs, nb, 1).ravel(), np.tile(dofs, (1, nb)).ravel())), shape=(N, N)).tocsr()


def test_linear_elasticity_3d_constant_tensor():
    """test_integration_linear_elasticity.py:205-322: constant Hooke tensor => HMM matrix == plain FEM matrix (rel 1e-4)."""
    msh = mesh.create_box([(0, 0, 0), (1.0, 0.2, 0.2)], (4, 2, 2))
    mic = mesh.create_unit_cube(3, 3, 3)
    g = 0.4 * 0.2**2
    h = with_oracle(hmm.LinearElasticityHMM(msh, lambda x, y: hmm.Lame(1.25, 1.0), lambda x: np.array([0.0, 0.0, -g]),
                                            mic, 1.0, petsc_options_cell_problem={"ksp_atol": 1e-9}))
    clamp = fem.locate_dofs_topological(h.function_space, 2,
                                        fem.locate_entities_boundary(msh, 2, lambda x: np.isclose(x[0], 0)))
    h.set_boundary_conditions(fem.dirichletbc(np.zeros(3), clamp, h.function_space))
    u = h.solve()
    K = _plain_elasticity_matrix(msh, 1.25, 1.0)
    assert np.linalg.norm((K - h._A).toarray()) / np.linalg.norm(K.toarray()) < 1e-12
    # and the solution equals the plain FEM solution
    b = fem.assemble_load_vector(h.function_space, lambda x: np.array([0.0, 0.0, -g]))
    idx, val = h._bcs[0].unrolled()
    import scipy.sparse.linalg as spla

    free = np.setdiff1d(np.arange(K.shape[0]), idx)
    uf = np.zeros(K.shape[0])
    uf[free] = spla.spsolve(K[free][:, free].tocsc(), b[free])
    assert np.abs(uf - u.x.array).max() < 1e-10 * np.abs(uf).max()
    assert u.x.array.reshape(-1, 3)[:, 2].min() < 0  # the beam bends downwards


def test_custom_function_valued_bc_and_reassembly_cache():
    """test_integration_poisson.py:322-395 (non-zero function-valued Dirichlet data) + hmm.py:150, 287, 300-301."""
    msh, mic = mesh.create_unit_square(6, 6), mesh.create_unit_square(4, 4)
    h = with_oracle(hmm.PoissonHMM(msh, lambda x, y: 1.0 + 0 * y[0], lambda x: 0.0, mic, 0.01))
    V = h.function_space
    gfun = fem.Function(V)
    gfun.interpolate(lambda x: 1.0 + 2.0 * x[0] - x[1])
    h.set_boundary_conditions(fem.dirichletbc(gfun, hmm._box_boundary_nodes(msh), V))
    u = h.solve()
    assert np.allclose(u.x.array, gfun.x.array, atol=1e-12)  # harmonic (affine) data, A = 1: exact
    calls = []
    orig = h._plan.solve
    h._plan.solve = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    h.solve()
    assert not calls  # cached until the boundary conditions change
    h.set_boundary_conditions(h._bcs)
    h.solve()
    assert calls


def test_constructor_checks():
    with pytest.raises(ValueError):
        hmm.PoissonHMM(mesh.create_unit_square(2, 2), lambda x, y: 1.0, lambda x: 1.0, mesh.create_unit_cube(2, 2, 2), 0.1)
    h = hmm.PoissonStratifiedHMM(mesh.create_unit_square(2, 2), lambda x, y: 1.0, lambda x: 1.0,
                                 mesh.create_unit_square(4, 4), 0.1, lambda x: np.ones((2, 1)))
    with pytest.raises(ValueError):
        h._stratification(np.arange(2))


def test_quadrature_degree_policy():
    """Default policy of BaseHMM(quadrature_degree=None): what UFL would estimate for the reference's coefficient families
    (hmm.py:190-198 + fem.form at :644-647)."""
    from hommx_amd import workloads as W

    msh, mic = mesh.create_unit_square(2, 2), mesh.create_unit_square(8, 8)
    mk = lambda A, **kw: hmm.PoissonHMM(msh, A, lambda x: 1.0, mic, 0.01, **kw)
    cells = np.arange(msh.num_cells)
    # smooth coefficients of the reference's tests -> degree 3 (6-point rule)
    for A in (lambda x, y: 1.0 / (2.0 + np.cos(2 * np.pi * y[0])), lambda x, y: 0.33 + 0.15 * (np.sin(2 * np.pi * x[0]) + np.sin(2 * np.pi * y[0]))):
        h = mk(A)
        coef, _ = h._element_means(cells)
        assert h.quadrature_degree_used == 3
        assert np.allclose(coef[0], O.sample_coefficient(A, msh.cell_midpoints()[0], 2, 8, 3), rtol=0, atol=1e-15)
    # conditionals between constants -> centroid rule, also when the interface cuts through elements (wrapped disc, inclusion.py:107-118)
    for A in (lambda x, y: np.where(np.cos(2 * np.pi * y[0]) < 0, 5.0, 0.05), lambda x, y: np.where(W.wrapped_disc(y[0], y[1]), 0.001 * (1 + 9 * x[0]), 0.1)):
        h = mk(A)
        coef, _ = h._element_means(cells)
        assert h.quadrature_degree_used == 0
        assert np.array_equal(coef[1], O.sample_coefficient(A, msh.cell_midpoints()[1], 2, 8, 0))
    # an explicit degree wins; TwoPhase is piecewise constant by construction
    assert mk(lambda x, y: 1.0 + y[0], quadrature_degree=2)._element_means(cells) is not None
    tp = hmm.TwoPhase(lambda y: y[0] < 0.5, lambda x: 2.0, lambda x: 1.0)
    assert mk(tp).quadrature_degree_used == 0
    with pytest.raises(ValueError):
        mk(tp, quadrature_degree=3)


def test_separable_host_stream_and_asymmetric_matrix_coefficient():
    """hmm.Separable: the documented host equivalent of the device sampler agrees with the generic callable path to rounding;
    a non-symmetric matrix-valued A is refused (the Schur form equals the reference's energy functional only for symmetric A)."""
    mic = mesh.create_unit_square(6, 6)
    bary, w = hmm.micro_quadrature(2, 3)
    yq = np.einsum("qa,eak->eqk", bary, mic.cell_vertices())
    c = np.array([[0.2, 0.3, 0.0], [0.7, 0.1, 0.0], [0.5, 0.9, 0.0]])
    for fam in ("affine", "reciprocal"):
        co = hmm.Separable(fam, lambda x: 2.0 + x[0], lambda x: 0.5 + 0.1 * x[1], lambda y: np.cos(2 * np.pi * y[0]) * np.sin(2 * np.pi * y[1]))
        stream = co.host_stream(co.params(c), co.table(yq, w), w)
        gen = np.stack([np.tensordot(w, np.asarray(co(c[k], yq.reshape(-1, 2).T)).reshape(yq.shape[:2]), axes=([0], [1])) for k in range(3)])
        assert np.abs(stream - gen).max() < 1e-14
    with pytest.raises(ValueError):
        hmm.Separable("cubic", None, None, None)
    msh = mesh.create_unit_square(2, 2)
    Aasym = lambda x, y: np.broadcast_to(np.array([[2.0, 0.5], [0.1, 1.0]]), (y.shape[1], 2, 2))
    h = hmm.PoissonHMM(msh, Aasym, lambda x: 1.0, mic, 0.01)
    with pytest.raises(ValueError, match="symmetric"):
        h._element_means(np.arange(msh.num_cells))
    Asym = lambda x, y: np.broadcast_to(np.array([[2.0, 0.3], [0.3, 1.0]]), (y.shape[1], 2, 2))
    coef, kind = hmm.PoissonHMM(msh, Asym, lambda x: 1.0, mic, 0.01)._element_means(np.arange(msh.num_cells))
    assert kind == "poisson_matrix" and coef.shape == (8, 72, 3)


def test_stratification_rejects_a_broadcast_that_is_wrong_in_the_interior(caplog):
    """a Dtheta_transpose that happens to broadcast but is wrong for interior cells (here: it reduces over its
    argument when handed the whole batch, yet the first and the last cell coincide with the per-cell call) must not be
    accepted on the evidence of the two end cells; the reference only ever calls it per cell (hmm.py:756-757)."""
    import logging

    msh = mesh.create_unit_square(4, 4)

    def dtheta(x):  # batch call: every cell gets the value of the batch's first cell in the (0, 1) entry, except the last one
        x = np.asarray(x, dtype=float)
        if x.ndim == 1:
            return np.array([[1.0, np.cos(x[0])], [0.0, 1.0]])
        off = np.full(x.shape[1], np.cos(x[0, 0]))
        off[-1] = np.cos(x[0, -1])
        return [[np.ones(x.shape[1]), off], [np.zeros(x.shape[1]), np.ones(x.shape[1])]]

    h = hmm.PoissonStratifiedHMM(msh, lambda x, y: 1.0 + 0 * y[0], lambda x: 1.0, mesh.create_unit_square(4, 4), 0.1, dtheta)
    cells = np.arange(msh.num_cells)
    with caplog.at_level(logging.DEBUG, logger="hommx_amd.hmm"):
        M = h._stratification(cells)
    c = msh.cell_midpoints()[cells]
    assert np.array_equal(M[:, 0, 1], np.cos(c[:, 0]))  # per-cell values everywhere
    assert any("broadcast call disagrees" in r.message for r in caplog.records)
    # a correct vectorised callable still takes the one-call path, bit for bit
    good = lambda x: [[1.0 + 0 * x[0], np.cos(x[0])], [0 * x[0], 1.0 + 0 * x[0]]]
    h2 = hmm.PoissonStratifiedHMM(msh, lambda x, y: 1.0 + 0 * y[0], lambda x: 1.0, mesh.create_unit_square(4, 4), 0.1, good)
    assert np.array_equal(h2._stratification(cells)[:, 0, 1], np.cos(c[:, 0]))


def test_prepare_reserves_the_plan_workspace_ahead_of_solve(monkeypatch):
    """`reserve=True` / `prepare()` create the plan and allocate its workspace for this rank's macro cells before
    the first solve (stub plan: records the call)."""
    made = []

    class StubPlan(OraclePlan):
        def __init__(self, dim, n, kind, device=0):
            super().__init__(dim, n, kind)
            self.device, self.reserved = device, []
            made.append(self)

        def reserve(self, n_cells):
            self.reserved.append(n_cells)

    monkeypatch.setattr(hmm, "MicroCellPlan", StubPlan)
    msh, mic = mesh.create_unit_square(3, 3), mesh.create_unit_square(4, 4)
    h = hmm.PoissonHMM(msh, lambda x, y: 1.0 + 0 * y[0], lambda x: 1.0, mic, 0.01, device=0)
    assert not made                                   # lazily by default: nothing happens at construction time
    h.prepare()
    assert len(made) == 1 and made[0].reserved == [msh.num_cells]
    h.solve()
    assert len(made) == 1 and made[0].reserved == [msh.num_cells]   # the solve reuses the plan and allocates nothing more
    tp = hmm.TwoPhase(lambda y: y[0] < 0.5, lambda x: 2.0 + x[0], lambda x: 1.0)
    h2 = hmm.PoissonHMM(msh, tp, lambda x: 1.0, mic, 0.01, device=0, reserve=True)
    assert len(made) == 2 and made[1].reserved == [msh.num_cells] and h2._plan is made[1]
