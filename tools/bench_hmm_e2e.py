"""End-to-end PoissonHMM.solve() on the C2 configuration (64x64 macro, 32x32 micro, inclusion), wall-clock split:
coefficient given as a generic callable A(x, y) (sampled on the host, streamed to the GPU) vs as hmm.TwoPhase (sampled on
the device).  python tools/bench_hmm_e2e.py [macro] [micro]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import fem, hmm, mesh, workloads as W

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ind = lambda y: W.wrapped_disc(y[0], y[1])
a_in = lambda x: 0.001 * (1.0 + 9.0 * x[0])
gen = lambda x, y: np.where(ind(y), a_in(x), 0.1)
tp = hmm.TwoPhase(ind, a_in, lambda x: 0.1)
sols = []
for name, A in (("callable", gen), ("TwoPhase", tp), ("callable", gen), ("TwoPhase", tp)):
    t0 = time.perf_counter()
    h = hmm.PoissonHMM(mesh.create_unit_square(nx, nx), A, lambda x: 1.0, mesh.create_unit_square(n, n), 2.0**-8)
    V = h.function_space
    left = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 0.0))
    right = fem.locate_dofs_geometrical(V, lambda x: np.isclose(x[0], 1.0))
    h.set_boundary_conditions([fem.dirichletbc(1.0, left, V), fem.dirichletbc(0.0, right, V)])
    t1 = time.perf_counter()
    h._assemble_stiffness()
    t2 = time.perf_counter()
    u = h.solve()
    t3 = time.perf_counter()
    sols.append(u.x.array.copy())
    print(f"{name:9s} setup {t1 - t0:6.3f} s | micro problems + macro assembly {t2 - t1:6.3f} s | macro BCs + solve {t3 - t2:6.3f} s"
          f" | total {t3 - t0:6.3f} s | {h._msh.num_cells} cells, bad {int((h.cell_info != 0).sum())}")
print("max |u_callable - u_twophase| =", float(np.abs(sols[2] - sols[3]).max()), " u range", float(sols[3].min()), float(sols[3].max()))
