import sys, time; sys.path.insert(0, '.')
import numpy as np
from oracle import hommx_oracle as O
from hommx_amd import MicroCellPlan
rng = np.random.default_rng(0)
ok = True
for n in (3, 4, 5, 8, 15, 16, 17, 24, 31, 32):
    plan = MicroCellPlan(2, n, "poisson")
    nc = 6
    coef = rng.uniform(0.05, 5.0, size=(nc, 2*n*n))
    M = np.eye(2)[None] + 0.4*rng.standard_normal((nc, 2, 2))
    for MM in (None, M):
        A, info = plan.solve(coef, MM, return_info=True)
        ref = O.effective_tensor_batch("poisson", 2, n, coef, MM)
        err = np.abs(A-ref).max()/np.abs(ref).max()
        print(n, plan.kernel, "M" if MM is not None else "-", "relerr %.2e" % err, info.tolist())
        ok &= err < 1e-10
print("ALL OK" if ok else "FAIL")
sys.exit(0 if ok else 1)
