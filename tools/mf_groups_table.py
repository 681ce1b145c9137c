import os, sys
sys.path.insert(0, "/root/repo")
os.environ["HOMMX_MF_VERBOSE"] = "1"
os.environ["HOMMX_MF_MIN_B"] = "1"
from hommx_amd import MicroCellPlan
for kind, dim, n in (("poisson", 2, 64), ("poisson", 2, 128), ("elasticity", 2, 32), ("elasticity", 2, 64), ("poisson", 3, 8), ("poisson", 3, 16), ("elasticity", 3, 8), ("elasticity", 3, 5)):
    print(f"=== {kind} {dim}D n={n}", file=sys.stderr, flush=True)
    p = MicroCellPlan(dim, n, kind, flags=1)
    print(p.kernel, p.flops_per_solve, file=sys.stderr, flush=True)
