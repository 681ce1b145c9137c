"""Dev tool: time hommx_solve_batch_correctors (host entry: coefficient stream in, tensors + correctors out) on one plan.
    python tools/bench_correctors.py [dim n kind cells]     (HOMMX_MF_CORR=0: plane elimination for multifrontal plans)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import MicroCellPlan

dim, n, kind, nc = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else (3, 16, "elasticity", 128)
p = MicroCellPlan(dim, n, kind)
rng = np.random.default_rng(0)
shape = (nc, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
coef = rng.uniform(0.5, 2.5, size=shape)
for rep in range(3):
    t0 = time.perf_counter()
    A, chi, info = p.solve(coef, None, return_info=True, return_correctors=True)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    A2 = p.solve(coef, None)
    dt2 = time.perf_counter() - t0
    print(f"{dim}D {kind} n={n} kernel={p.kernel} cells={nc}: tensors + correctors {dt*1e3:9.1f} ms = {nc/dt:9.1f} solves/s | tensors only "
          f"{dt2*1e3:9.1f} ms = {nc/dt2:9.1f} solves/s | bad={int((info != 0).sum())}  max|A - A2|/|A| = {np.abs(A - A2).max()/np.abs(A2).max():.1e}")
