"""Dev tool: the generic host entry hommx_solve_batch (pageable coefficient stream in, tensors out) on the C4 workload.
    python tools/bench_host_stream.py [cells]     (HOMMX_NO_H2D_OVERLAP=1: one copy in front of the kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import MicroCellPlan

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4320
p = MicroCellPlan(3, 16, "elasticity")
rng = np.random.default_rng(0)
coef = rng.uniform(0.5, 2.5, size=(nc, p.n_el, 2))
p.reserve(nc)
for rep in range(3):
    t0 = time.perf_counter()
    A, info = p.solve(coef, None, return_info=True)
    dt = time.perf_counter() - t0
    print(f"host stream, 3D elasticity 16^3 kernel={p.kernel} cells={nc}: {dt*1e3:9.1f} ms = {nc/dt:9.1f} solves/s  bad={int((info != 0).sum())}  "
          f"checksum {float(A.sum()):.12e}")
