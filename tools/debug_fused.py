"""Dev tool: per-size parity of the fused 2D kernel against the oracle (prints the error of every size, both with and without M)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import MicroCellPlan
from oracle import hommx_oracle as O

rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1:]] or [32, 16, 31, 17, 24, 20, 3, 4, 5, 7, 8, 15]
for n in sizes:
    p = MicroCellPlan(2, n, "poisson")
    coef = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(3, 2 * n * n)))
    M = np.eye(2)[None] + 0.4 * rng.standard_normal((3, 2, 2))
    for MM in (None, M):
        A, info = p.solve(coef, MM, return_info=True)
        ref = O.effective_tensor_batch("poisson", 2, n, coef, MM)
        err = np.abs(A - ref).max() / np.abs(ref).max()
        print(f"n={n:2d} M={'y' if MM is not None else 'n'} err={err:.2e} info={info.tolist()}", flush=True)
        if not (err < 1e-10):
            print("  got", A[0].ravel(), "\n  ref", ref[0].ravel())
