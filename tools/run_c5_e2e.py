"""End-to-end LinearElasticityStratifiedHMM.solve() of BASELINE configuration 5 on one MI355X: 32 x 16 x 8 macro box of the beam
1 x 0.4 x 0.1 (24,576 tetrahedra), 16^3 micro cells, rotated-fibre Hooke tensor (examples/linear_elasticity/rotated_fibers.py:23-76,
README.md:171-185), load f = (0, 0, -0.05 (W/L)^2), clamped at x0 = 0 (:102-115).  Wall-clock split of the solver class.
    python tools/run_c5_e2e.py [nx ny nz] [n_micro] [--reserve]
--reserve: the solver is built with reserve=True (plan + device workspace allocated in the constructor, hommx_plan_reserve: up to 128 GB)
instead of inside the first solve (64 GB): the hipMalloc moves from "micro problems" to "setup".  Run once with and once without."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import fem, hmm, mesh, workloads as W

reserve = "--reserve" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = tuple(int(a) for a in argv[0:3]) if len(argv) >= 3 else (32, 16, 8)
n = int(argv[3]) if len(argv) > 3 else 16
msh = mesh.create_box([(0, 0, 0), (1.0, 0.4, 0.1)], shape)
A = hmm.TwoPhase(lambda y: W.wrapped_disc(y[1], y[2]), lambda x: hmm.Lame(1.0 + 0 * x[0], 100.0 + 0 * x[0]),
                 lambda x: hmm.Lame(1.0 + 0 * x[0], 0.001 + 0 * x[0]))
def Dtheta_transpose(x):
    x = np.asarray(x, float)
    if x.ndim == 1:
        return W.c5_theta_transpose(x[None, :3])[0]
    M = W.c5_theta_transpose(x.T)           # [N, 3, 3]
    return np.moveaxis(M, 0, -1)            # [3, 3, N]: the broadcast form _stratification accepts
t0 = time.perf_counter()
h = hmm.LinearElasticityStratifiedHMM(msh, A, lambda x: np.array([0.0, 0.0, -0.05 * 0.4**2]), mesh.create_unit_cube(n, n, n), 2.0**-5,
                                      Dtheta_transpose, reserve=reserve)
V = h.function_space
clamp = fem.locate_dofs_topological(V, 2, fem.locate_entities_boundary(msh, 2, lambda x: np.isclose(x[0], 0)))
h.set_boundary_conditions(fem.dirichletbc(np.zeros(3), clamp, V))
t1 = time.perf_counter()
h._assemble_stiffness()
t2 = time.perf_counter()
u = h.solve()
t3 = time.perf_counter()
U = u.x.array.reshape(-1, 3)
print(f"reserve={reserve}  cells {msh.num_cells}  micro {n}^3  kernel {h._plan.kernel}  setup {t1 - t0:.2f} s | micro problems + macro assembly {t2 - t1:.2f} s | "
      f"macro BCs + solve {t3 - t2:.2f} s | total {t3 - t0:.2f} s | bad cells {int((h.cell_info != 0).sum())}")
print(f"tip deflection u_z(x0 = 1) mean {U[np.isclose(msh.geometry.x[:, 0], 1.0), 2].mean():.6e}  max |u| {np.abs(U).max():.6e}  "
      f"C_H[0] diag {np.round(np.diag(h.effective_tensors[0]), 4).tolist()}")
