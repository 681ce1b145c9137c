"""The reference's own acceptance test for boundary data, homogenisation and stratification: the HMM solution against a plain P1 solve of the
heterogeneous problem on a 1024 x 1024 mesh that resolves eps = 2^-6, relative L2 error under the reference's thresholds

    test_integration_poisson.py:322-395   custom (function-valued) Dirichlet data, A = 1.1 + x0 + sin(2 pi y0)          < 8e-4
    test_integration_poisson.py:398-478   the same without micro structure, A = 1.1 + x0                                < 5e-4
    test_integration_poisson.py:481-572   stratified, y = theta(x) / eps with the reference's theta and its D theta^T   < 1e-2
    test_integration_poisson.py:243-294   3D: 6^3 macro, 6^3 micro cells, eps = 1/8, fine mesh 64^3                     < 5e-2
    test_integration_linear_elasticity.py:62-171   2D beam 1 x 0.2 clamped at x = 0 under its weight, mu = 5 + 4.5 sin(2 pi y0),
                                          lambda = 1.25, 40 x 12 macro, 10 x 10 micro cells, eps = 2^-6, fine mesh 800 x 240    < 4e-2

with the same meshes (2D: 15 x 15 macro, 15 x 15 micro), the same eps, the same data.  One more case of OUR choosing pins the coefficient
family of the headline configuration, which none of the reference's tests covers (examples/inclusion.py:107-118, no asserted numbers): a
wrapped-disc inclusion with an x-dependent inner value, eps = 1/16, 16 x 16 macro and 32 x 32 micro cells (the fused 2D kernel, two-phase device
sampler) against the same kind of fine-scale solve -- observed 2.2e-3, asserted < 4e-3 (our heuristic, not the reference's).  The fine-scale solve is tests/fine_fem.py (NumPy / SciPy,
multigrid-preconditioned CG): an independent discretisation of the ORIGINAL problem, sharing no code with the solver classes or the oracle.
CPU: the solver classes with the oracle standing in for the GPU plan; GPU: the HIP path."""
import functools

import numpy as np
import pytest

import fine_fem
from hommx_amd import fem, hmm, mesh

EPS = 2.0**-6
N_FINE = 1024
TF = 0.2  # theta_factor of the reference's test_stratified


def A_osc(x, y):
    return 1.1 + x[0] + np.sin(2 * np.pi * y[0])


def A_smooth(x, y):
    return 1.1 + x[0] + 0.0 * y[0]


def g_bc(x):
    return 1 + x[0] ** 2 + x[1] ** 2


def theta(x):
    fac = TF * np.cos(np.pi / 2 * x[1]) * np.cos(np.pi / 2 * x[0])
    return np.stack([x[0] - fac * x[1], x[1] + fac * x[0]])


def Dtheta_t(x):
    """The reference's `Dtheta` argument (test_integration_poisson.py:499-508): entry [i][j] = d theta_j / d x_i."""
    a0, a1 = np.pi / 2 * x[0], np.pi / 2 * x[1]
    f = TF * np.cos(a0) * np.cos(a1)
    df0 = -TF * (np.pi / 2) * np.sin(a0) * np.cos(a1)
    df1 = -TF * (np.pi / 2) * np.cos(a0) * np.sin(a1)
    return np.array([[1 - x[1] * df0, f + x[0] * df0], [-f - x[1] * df1, 1 + x[0] * df1]])


@functools.lru_cache(maxsize=None)
def fine_solution(case: str):
    if case == "osc":
        return fine_fem.solve_fine(N_FINE, lambda x: A_osc(x, x / EPS), 1.0, g_bc)[0]
    if case == "smooth":
        return fine_fem.solve_fine(N_FINE, lambda x: A_smooth(x, x / EPS), 1.0, g_bc)[0]
    return fine_fem.solve_fine(N_FINE, lambda x: A_osc(x, theta(x) / EPS), 1.0, lambda x: 0.0 * x[0])[0]


def relative_error(h, u, case):
    """calc_l2_error(u_hmm, I u_ref) / calc_l2_norm(I u_ref) with I = interpolation of the fine solution into the macro space."""
    V = h.function_space
    ref_nodal = fine_fem.sample_p1(fine_solution(case), V.tabulate_dof_coordinates()[:, :2])
    return fine_fem.relative_l2_error_p1(V.mesh, u.x.array, ref_nodal)


def build(case, plan_hook):
    msh, mic = mesh.create_unit_square(15, 15), mesh.create_unit_square(15, 15)
    one = lambda x: 1.0
    if case == "strat":
        h = hmm.PoissonStratifiedHMM(msh, A_osc, one, mic, EPS, Dtheta_t, petsc_options_cell_problem={"ksp_type": "gmres", "pc_type": "none"})
    else:
        h = hmm.PoissonHMM(msh, A_osc if case == "osc" else A_smooth, one, mic, EPS, petsc_options_cell_problem={"ksp_atol": 1e-9})
    h = plan_hook(h)
    V = h.function_space
    facets = fem.locate_entities_boundary(msh, 1, lambda x: np.isclose(x[0], 0) | np.isclose(x[0], 1) | np.isclose(x[1], 0) | np.isclose(x[1], 1))
    dofs = fem.locate_dofs_topological(V, 1, facets)
    if case == "strat":
        h.set_boundary_conditions(fem.dirichletbc(0.0, dofs, V))
    else:
        gfun = fem.Function(V)
        gfun.interpolate(g_bc)
        h.set_boundary_conditions(fem.dirichletbc(gfun, dofs, V))
    return h


CASES = [("osc", 8e-4), ("smooth", 5e-4), ("strat", 1e-2)]

EPS_3D = 1.0 / 2**3


@functools.lru_cache(maxsize=None)
def fine_solution_3d():
    return fine_fem.solve_fine_3d(64, lambda x: A_osc(x, x / EPS_3D), 1.0)[0]


def run_3d(plan_hook):
    msh, mic = mesh.create_unit_cube(6, 6, 6), mesh.create_unit_cube(6, 6, 6)
    h = plan_hook(hmm.PoissonHMM(msh, A_osc, lambda x: 1.0, mic, EPS_3D, petsc_options_cell_problem={"ksp_atol": 1e-9}))
    V = h.function_space
    on_box = lambda x: (np.isclose(x[0], 0) | np.isclose(x[0], 1) | np.isclose(x[1], 0) | np.isclose(x[1], 1) | np.isclose(x[2], 0)
                        | np.isclose(x[2], 1))
    h.set_boundary_conditions(fem.dirichletbc(0.0, fem.locate_dofs_topological(V, 2, fem.locate_entities_boundary(msh, 2, on_box)), V))
    u = h.solve()
    ref_nodal = fine_fem.sample_p1_3d(fine_solution_3d(), V.tabulate_dof_coordinates()[:, :3])
    return h, fine_fem.relative_l2_error_p1_3d(V.mesh, u.x.array, ref_nodal)


BEAM = (1.0, 0.2)
G_BEAM = 0.4 * (BEAM[1] / BEAM[0]) ** 2


@functools.lru_cache(maxsize=None)
def fine_solution_beam():
    return fine_fem.solve_fine_elasticity_2d(800, 240, BEAM[0], BEAM[1], lambda x: 1.25 + 0.0 * x[0],
                                             lambda x: 5.0 + 4.5 * np.sin(2 * np.pi * x[0] / EPS), (0.0, -G_BEAM))


def run_beam(plan_hook):
    msh = mesh.create_rectangle([(0, 0), BEAM], (40, 12))
    # lambda = 1.25, mu = 5 + 4.5 sin 2 pi y0 as a Separable coefficient: sampled on the device by the GPU plan (hommx_solve_batch_separable),
    # through the generic callable path by the oracle plan -- the same element means either way
    A = hmm.Separable("affine", lambda x: hmm.Lame(1.25 + 0.0 * x[0], 5.0 + 0.0 * x[0]), lambda x: hmm.Lame(0.0 * x[0], 4.5 + 0.0 * x[0]),
                      lambda y: np.sin(2 * np.pi * y[0]))
    h = plan_hook(hmm.LinearElasticityHMM(msh, A, lambda x: np.array([0.0, -G_BEAM]), mesh.create_unit_square(10, 10), EPS,
                                          petsc_options_cell_problem={"ksp_atol": 1e-9}))
    V = h.function_space
    clamp = fem.locate_dofs_topological(V, 1, fem.locate_entities_boundary(msh, 1, lambda x: np.isclose(x[0], 0)))
    h.set_boundary_conditions(fem.dirichletbc(np.zeros(2), clamp, V))
    u = h.solve().x.array.reshape(-1, 2)
    ref = fine_fem.sample_p1_rect(fine_solution_beam(), msh.geometry.x[:, :2], *BEAM)
    num = sum(fine_fem.relative_l2_error_p1(msh, u[:, c], ref[:, c]) ** 2 * _sq(msh, ref[:, c]) for c in range(2))
    den = sum(_sq(msh, ref[:, c]) for c in range(2))
    return h, np.sqrt(num / den)


def _sq(msh, w):
    cv, vol = msh.cells, msh.cell_volumes()
    a, b, c = w[cv[:, 0]], w[cv[:, 1]], w[cv[:, 2]]
    return float(np.sum(vol / 6.0 * (a * a + b * b + c * c + a * b + b * c + a * c)))


EPS_INC = 1.0 / 16


def _inclusion():
    from hommx_amd import workloads as W

    ind = lambda y: W.wrapped_disc(y[0], y[1])
    a_in = lambda x: 0.05 * (1.0 + 9.0 * x[0])
    return ind, a_in


@functools.lru_cache(maxsize=None)
def fine_solution_inclusion():
    ind, a_in = _inclusion()
    return fine_fem.solve_fine(N_FINE, lambda x: np.where(ind(x / EPS_INC), a_in(x), 1.0), 1.0, lambda x: 0.0 * x[0])[0]


def run_inclusion(plan_hook):
    ind, a_in = _inclusion()
    msh, mic = mesh.create_unit_square(16, 16), mesh.create_unit_square(32, 32)
    h = plan_hook(hmm.PoissonHMM(msh, hmm.TwoPhase(ind, a_in, lambda x: 1.0 + 0.0 * x[0]), lambda x: 1.0, mic, EPS_INC))
    V = h.function_space
    on_box = lambda x: np.isclose(x[0], 0) | np.isclose(x[0], 1) | np.isclose(x[1], 0) | np.isclose(x[1], 1)
    h.set_boundary_conditions(fem.dirichletbc(0.0, fem.locate_dofs_topological(V, 1, fem.locate_entities_boundary(msh, 1, on_box)), V))
    u = h.solve()
    ref = fine_fem.sample_p1(fine_solution_inclusion(), V.tabulate_dof_coordinates()[:, :2])
    return h, fine_fem.relative_l2_error_p1(msh, u.x.array, ref)


def test_fine_solver_against_a_manufactured_solution():
    """The checker itself: constant coefficient, u = 1 + x^2 + y^2 solves -div(grad u) = -4 exactly at the nodes of this stencil."""
    u, its = fine_fem.solve_fine(64, lambda x: 1.0 + 0.0 * x[0], -4.0, g_bc)
    xn, yn = np.meshgrid(np.arange(65) / 64, np.arange(65) / 64, indexing="ij")
    assert np.abs(u - (1 + xn**2 + yn**2)).max() < 1e-9 and its < 30
    # variable coefficient, two resolutions: second-order convergence at a fixed point
    A = lambda x: 1.0 + 0.5 * np.sin(2 * np.pi * x[0]) * np.cos(2 * np.pi * x[1])
    vals = [fine_fem.solve_fine(N, A, 1.0, lambda x: 0.0 * x[0])[0][N // 2, N // 4] for N in (32, 64, 128)]
    assert abs(vals[2] - vals[1]) < 0.35 * abs(vals[1] - vals[0])


@pytest.mark.parametrize("case,tol", CASES)
def test_hmm_vs_fine_scale_fem_cpu(case, tol):
    from test_hmm_host import with_oracle

    h = build(case, with_oracle)
    u = h.solve()
    err = relative_error(h, u, case)
    assert err < tol, err


def test_hmm_3d_vs_fine_scale_fem_cpu():
    from test_hmm_host import with_oracle

    _, err = run_3d(with_oracle)
    assert err < 0.05, err


def test_inclusion_vs_fine_scale_fem_cpu():
    from test_hmm_host import with_oracle

    _, err = run_inclusion(with_oracle)
    assert err < 4e-3, err


@pytest.mark.gpu
def test_inclusion_vs_fine_scale_fem_gpu():
    """Coefficient family of C2 through the fused 2D kernel (n = 32) with the two-phase sampler on the device."""
    h, err = run_inclusion(lambda h: h)
    assert h._plan.kernel == "fused2d" and not h.cell_info.any()
    assert err < 4e-3, err


def test_elasticity_beam_vs_fine_scale_fem_cpu():
    from test_hmm_host import with_oracle

    _, err = run_beam(with_oracle)
    assert err < 0.04, err


@pytest.mark.gpu
def test_elasticity_beam_vs_fine_scale_fem_gpu():
    """10 x 10 micro cells, two components: plane block b = 20, bordered one-wave-per-cell kernel."""
    h, err = run_beam(lambda h: h)
    assert not h.cell_info.any()
    assert err < 0.04, err


@pytest.mark.gpu
def test_hmm_3d_vs_fine_scale_fem_gpu():
    """6^3 micro cells: plane block b = 36, the one-wave-per-cell register kernel (csrc/small_wave.h)."""
    h, err = run_3d(lambda h: h)
    assert not h.cell_info.any()
    assert err < 0.05, err


@pytest.mark.gpu
@pytest.mark.parametrize("case,tol", CASES)
def test_hmm_vs_fine_scale_fem_gpu(case, tol):
    h = build(case, lambda h: h)
    u = h.solve()
    assert not h.cell_info.any()
    err = relative_error(h, u, case)
    assert err < tol, err
