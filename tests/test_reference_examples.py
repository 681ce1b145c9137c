"""The two numbers the reference holds for the coefficient families of the BASELINE configurations.

examples/diffusion/inclusion.py:16 (``vmax=2.3550577798756827``) and examples/diffusion/laminate.py:16 (``1.3934988798958294``) are the
colour-bar maxima of the authors' fine-scale solutions: P1 on ``create_unit_square(1024, 1024)``, A = ``conditional`` of theta(x) / eps,
eps = 2^-5, f = 1, u = 1 / 0 on x0 = 0 / 1, PETSc CG + GAMG at its default rtol 1e-5 (inclusion.py:107-161, laminate.py:101-144).  They
depend on exactly the conventions the reference's tests leave unpinned: the direction of the mesh diagonal and where a ``conditional`` is
sampled (UFL: degree 0, the triangle's centroid).

Observed with tests/fine_fem.py::solve_fine_darcy (independent NumPy / SciPy solver, CG to 1e-11):

  inclusion, right diagonal (DOLFINx default), centroid sampling   max u = 2.3550938289   rel. diff to the reference's number  +1.5e-5
  inclusion, LEFT diagonal                                          max u = 2.3566219100                                          +6.6e-4
  -> the number is reproduced to what rtol 1e-5 allows, and it tells the two triangulations apart by a factor of 40: the right-diagonal
     split and the centroid sampling of a conditional are PINNED by a reference-held value.
  laminate, right diagonal   N = 512: 1.2510891   N = 1024: 1.3583400   N = 2048: 1.3991515      reference's number 1.3934989
  laminate, left diagonal    N = 1024: 1.3428022
  -> the FINE solve of the script as committed does not give the number it holds (-2.5e-2 at its own N = 1024; the layers of theta0 =
     x1 - sin(2 pi x0) are 1/410 thick where the wave is steepest, 2.5 elements, and max u still moves by 3 % from N = 1024 to 2048).
     The number is the maximum of the script's HMM solution instead (laminate.py:177-203: PoissonStratifiedHMM, 30 x 30 macro and micro
     cells, Darcy data; both plots share the colour bar):

  laminate, PoissonStratifiedHMM 30 x 30 / 30 x 30 (this package, oracle plan or GPU)   max u = 1.3935001295   rel. diff  +9.0e-7
  -> the WHOLE pipeline -- centroid sampling, stratified cell problems, S_loc scaling, macro assembly, Dirichlet lifting, macro solve --
     reproduces a reference-held value to what the reference's own Krylov tolerances (rtol 1e-5) allow.
  inclusion, same HMM set-up: max u = 2.3273691 (1.2 % below the fine-scale maximum: that number is NOT the HMM one).

The HMM side of both examples runs on the CPU with the oracle plan and on the GPU; the relative L2 distances to the fine-scale solutions
are our heuristics (the reference asserts nothing for these scripts): inclusion 7.6e-3 observed, laminate 1.9e-2."""
import functools

import numpy as np
import pytest

import fine_fem
from hommx_amd import fem, hmm, mesh

EPS = 2.0**-5
REF_MAX_INCLUSION = 2.3550577798756827  # examples/diffusion/inclusion.py:16
REF_MAX_LAMINATE = 1.3934988798958294   # examples/diffusion/laminate.py:16


def disc(y0, y1):
    """ufl_circle_indicator of inclusion.py:107-114."""
    dx = np.arccos(np.cos(2 * np.pi * (y0 - 0.5)))
    dy = np.arccos(np.cos(2 * np.pi * (y1 - 0.5)))
    return dx**2 + dy**2 < (2 * np.pi) ** 2 * 0.25**2


def theta_inclusion(x):
    """inclusion.py:121-125."""
    return np.stack([x[0] + 0.5 * np.sin(2 * np.pi * x[1]), x[1]])


def dtheta_t_inclusion(x):
    """inclusion.py:128-134: transpose of [[1, pi cos 2 pi x1], [0, 1]], i.e. entry [i][j] = d theta_j / d x_i."""
    return np.array([[1.0 + 0 * x[0], 0.0 * x[0]], [np.pi * np.cos(2 * np.pi * x[1]), 1.0 + 0 * x[0]]])


def A_fine_inclusion(x):
    y = theta_inclusion(x) / EPS
    return np.where(disc(y[0], y[1]), 0.001, 0.1)  # inclusion.py:117-118


def A_fine_laminate(x):
    """laminate.py:101-102 with theta0 = x1 - sin 2 pi x0 (:109-112)."""
    return np.where(np.cos(2 * np.pi * (x[1] - np.sin(2 * np.pi * x[0])) / EPS) < 0, 5.0, 0.05)


@functools.lru_cache(maxsize=None)
def fine(case: str, diagonal: str = "right", N: int = 1024):
    return fine_fem.solve_fine_darcy(N, A_fine_inclusion if case == "inclusion" else A_fine_laminate, 1.0, diagonal)[0]


def test_fine_darcy_solver_on_a_known_answer():
    """The checker itself: A = 1 gives u = 1 - x + x (1 - x) / 2, independent of y (natural conditions on x1 = 0, 1), exactly at the nodes."""
    for diag in ("right", "left"):
        u, its = fine_fem.solve_fine_darcy(64, lambda x: 1.0 + 0.0 * x[0], 1.0, diag)
        xn = np.arange(65)[:, None] / 64 + 0.0 * np.arange(65)[None, :]
        assert np.abs(u - (1 - xn + 0.5 * xn * (1 - xn))).max() < 1e-9 and its < 40


def test_inclusion_example_maximum_is_reproduced_and_tells_the_diagonals_apart():
    right = fine("inclusion", "right").max()
    left = fine("inclusion", "left").max()
    assert abs(right / REF_MAX_INCLUSION - 1.0) < 5e-5, right   # observed +1.5e-5: what PETSc's default rtol 1e-5 leaves
    assert abs(left / REF_MAX_INCLUSION - 1.0) > 3e-4, left     # observed +6.6e-4: the other diagonal is NOT what the authors ran


def test_laminate_example_fine_maximum_is_resolution_bound():
    """The committed script's fine solve (N = 1024) does not give the number it holds (module docstring): it is the HMM maximum."""
    right = fine("laminate", "right").max()
    assert abs(right - 1.3583400092) < 1e-8                      # our own regression value
    assert right < REF_MAX_LAMINATE < 1.3991515166               # bracketed by N = 1024 and N = 2048 (the latter observed once, 140 s)
    assert abs(right / REF_MAX_LAMINATE - 1.0) > 2e-2


def test_laminate_example_hmm_maximum_reproduces_the_reference_held_number_cpu():
    """laminate.py:16 is max(u_hmm) of laminate.py:177-203; the solver classes with the ORACLE standing in for the GPU plan reproduce it."""
    from test_hmm_host import with_oracle

    h = with_oracle(hmm_laminate())
    u = h.solve()
    assert h.quadrature_degree_used == 0 and not h.cell_info.any()
    assert abs(float(u.x.array.max()) / REF_MAX_LAMINATE - 1.0) < 1e-5, u.x.array.max()   # observed +9.0e-7
    err, _ = rel_err(h, u, "laminate")
    assert err < 3e-2, err                                                                # observed 1.9e-2 (heuristic, ours)


def darcy(h):
    V = h.function_space
    left = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0.0))
    right = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 1.0))
    h.set_boundary_conditions([fem.dirichletbc(1.0, left, V), fem.dirichletbc(0.0, right, V)])
    return h


def hmm_inclusion():
    """inclusion.py:197-211: 30 x 30 macro and micro cells, PoissonStratifiedHMM, eps plays no role."""
    A = hmm.TwoPhase(lambda y: disc(y[0], y[1]), lambda x: 0.001 + 0.0 * x[0], lambda x: 0.1 + 0.0 * x[0])
    return darcy(hmm.PoissonStratifiedHMM(mesh.create_unit_square(30, 30), A, lambda x: 1.0, mesh.create_unit_square(30, 30), 1e-5,
                                          dtheta_t_inclusion))


def hmm_laminate():
    """laminate.py:177-185 with the runnable square D theta^T: the script passes a 2 x 1 matrix, which hmm.py:762 cannot multiply with a
    2-vector gradient; the same layered medium in the square convention of the reference's test (test_integration_poisson.py:499-508) is
    theta = (x0, x1 - sin 2 pi x0) with the laminate in y1 (README.md:96-99)."""
    A = hmm.TwoPhase(lambda y: np.cos(2 * np.pi * y[1]) < 0, lambda x: 5.0 + 0.0 * x[0], lambda x: 0.05 + 0.0 * x[0])
    Dt = lambda x: np.array([[1.0 + 0 * x[0], -2 * np.pi * np.cos(2 * np.pi * x[0])], [0.0 * x[0], 1.0 + 0 * x[0]]])
    return darcy(hmm.PoissonStratifiedHMM(mesh.create_unit_square(30, 30), A, lambda x: 1.0, mesh.create_unit_square(30, 30), 1e-5, Dt))


def rel_err(h, u, case):
    V = h.function_space
    ref = fine_fem.sample_p1(fine(case), V.tabulate_dof_coordinates()[:, :2])
    return fine_fem.relative_l2_error_p1(V.mesh, u.x.array, ref), float(u.x.array.max())


@pytest.mark.gpu
def test_inclusion_example_hmm_vs_fine_scale_gpu():
    h = hmm_inclusion()
    u = h.solve()
    assert h._plan.kernel == "fused2d" and not h.cell_info.any() and h.quadrature_degree_used == 0
    err, umax = rel_err(h, u, "inclusion")
    assert err < 1.2e-2, (err, umax)                      # heuristic (ours; observed 7.6e-3): eps = 1/32 against the eps -> 0 limit
    assert abs(umax - 2.3273690632) < 1e-7, umax          # the oracle-plan value of the same set-up (module docstring)


@pytest.mark.gpu
def test_laminate_example_hmm_vs_fine_scale_gpu():
    h = hmm_laminate()
    u = h.solve()
    assert not h.cell_info.any()
    err, umax = rel_err(h, u, "laminate")
    assert abs(umax / REF_MAX_LAMINATE - 1.0) < 1e-5, umax   # laminate.py:16, the maximum of the reference's own HMM solution (+9.0e-7)
    assert err < 3e-2, (err, umax)                        # heuristic (ours; observed 1.9e-2): the fine solution is 3 % from converged
