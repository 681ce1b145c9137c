"""Dev tool: throughput of the blocked path across dimensions / kinds / sizes (random coefficients, device-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan

only_small = "--small" in sys.argv

dev = torch.device("cuda:0")
cases = [(2, 10, "elasticity", 0, 8192), (3, 6, "poisson", 0, 8192), (2, 16, "poisson_matrix", 0, 8192), (3, 8, "poisson", 0, 4096),
         (2, 32, "poisson", 1, 4096), (2, 32, "poisson_matrix", 0, 4096), (2, 32, "elasticity", 0, 4096), (2, 16, "elasticity", 0, 4096),
         (2, 64, "poisson", 0, 2048), (3, 8, "poisson", 0, 2048), (2, 64, "elasticity", 0, 2048), (2, 96, "poisson", 0, 2048),
         (2, 128, "poisson", 0, 2048), (3, 9, "poisson", 0, 1024), (3, 12, "poisson", 0, 1024), (3, 16, "poisson", 0, 1024),
         (3, 8, "elasticity", 0, 1024), (3, 12, "elasticity", 0, 1024), (3, 16, "elasticity", 0, 1024)]
if only_small:
    cases = cases[:4]
for a in sys.argv[1:]:  # --case=dim,n,kind,cells
    if a.startswith("--case="):
        d_, n_, k_, c_ = a[7:].split(",")
        cases = [(int(d_), int(n_), k_, 1, int(c_))]
for dim, n, kind, flags, nc in cases:
    p = MicroCellPlan(dim, n, kind, flags=flags)
    p.reserve(nc)  # workspace of the blocked family ahead of the timed calls (up to 128 GB instead of the 64 GB a first solve takes)
    shape = (nc, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
    g = torch.Generator(device="cpu").manual_seed(0)
    coef = torch.rand(shape, dtype=torch.float64, generator=g) * 2 + 0.5
    if kind == "poisson_matrix":  # SPD 2x2 in (xx, yy, xy) order: keep the off-diagonal small
        coef[..., -1] = 0.1
    coef = coef.to(dev)
    out = torch.empty(nc, p.t, p.t, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    b = p.n_comp and (1 if kind.startswith("poisson") else dim) * n ** (dim - 1)
    print(f"{dim}D {kind:15s} n={n:3d} b={b:4d} kernel={p.kernel:8s} cells={nc}: {dt*1e3:9.2f} ms  {nc/dt:12.1f} solves/s  "
          f"model {(6*(n-1)+2)*b**3*nc/dt/1e12:6.2f} TF/s  bad={int((info!=0).sum())}")
