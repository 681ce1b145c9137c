import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import MicroCellPlan
from oracle import hommx_oracle as O
rng = np.random.default_rng(5)
for n in (32, 24, 17):
    for lo, hi in ((1e-1, 1e1), (1e-3, 1e3), (1e-5, 1e5)):
        coef = np.exp(rng.uniform(np.log(lo), np.log(hi), size=(4, 2 * n * n)))
        M = np.eye(2)[None] + 0.3 * rng.standard_normal((4, 2, 2))
        p = MicroCellPlan(2, n, "poisson")
        A, info = p.solve(coef, M, return_info=True)
        ref = O.effective_tensor_batch("poisson", 2, n, coef, M)
        print(n, lo, hi, p.kernel, "relerr", np.abs(A - ref).max() / np.abs(ref).max(), "info", info.tolist())
