"""Dev tool: the multifrontal route (csrc/multifrontal.hip) against the oracle on small meshes, then its throughput at 16^3.
HOMMX_MF_MIN_B is read when a plan is created: run with HOMMX_MF_MIN_B=65 to force the route, =0 to switch it off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import MicroCellPlan

rng = np.random.default_rng(1)
if "--time-only" not in sys.argv:
    from oracle import hommx_oracle as O
    for kind, dim, n in (("elasticity", 3, 5), ("elasticity", 3, 6), ("poisson", 3, 9), ("elasticity", 3, 8), ("poisson", 3, 12), ("poisson_matrix", 3, 7)):
        p = MicroCellPlan(dim, n, kind)
        nc = 5
        shape = (nc, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
        coef = rng.uniform(0.5, 3.0, size=shape)
        if kind == "poisson_matrix":
            coef[..., 3:] *= 0.1
        M = np.eye(dim)[None] + 0.2 * rng.standard_normal((nc, dim, dim))
        A, info = p.solve(coef, M, return_info=True)
        if kind == "poisson_matrix":
            full = np.zeros((nc, p.n_el, 3, 3))
            for q, (i, j) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
                full[..., i, j] = full[..., j, i] = coef[..., q]
            ref = O.effective_tensor_batch("poisson", dim, n, full[:2], M[:2])
        else:
            ref = O.effective_tensor_batch(kind, dim, n, coef[:2], M[:2])
        print(f"{kind} {dim}D n={n} kernel={p.kernel} info={info.tolist()} rel err {np.abs(A[:2]-ref).max()/np.abs(ref).max():.2e}", flush=True)
import torch
dev = torch.device("cuda:0")
for nc in [int(a) for a in sys.argv[1:] if a.isdigit()] or [256]:
    p = MicroCellPlan(3, 16, "elasticity")
    p.reserve(nc)  # workspace of the blocked family ahead of the timed calls (up to 128 GB instead of the 64 GB a first solve takes)
    g = torch.Generator(device="cpu").manual_seed(0)
    coef = (torch.rand((nc, p.n_el, 2), dtype=torch.float64, generator=g) * 2 + 0.5).to(dev)
    out = torch.empty(nc, 6, 6, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"3D elasticity 16^3 kernel={p.kernel} cells={nc}: {dt*1e3:.1f} ms  {nc/dt:.1f} solves/s  bad={int((info!=0).sum())} codes={sorted(set(info.cpu().tolist()))}  sum={float(out.sum()):.12e}", flush=True)
