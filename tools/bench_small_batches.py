"""Dev tool: small, launch-bound batches of the nested-dissection route (20 repeated device-pointer calls with the same arguments; it
timed plain launches against a hipGraph replay, which was slower and is not in the tree: DESIGN.md 4.4).
    python tools/bench_small_batches.py dim n kind cells [cells ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan

dim, n, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
for nc in [int(a) for a in sys.argv[4:]]:
    p = MicroCellPlan(dim, n, kind)
    g = torch.Generator(device="cpu").manual_seed(0)
    shape = (nc, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
    coef = (torch.rand(shape, dtype=torch.float64, generator=g) * 2 + 0.5).to(dev)
    out = torch.empty(nc, p.t, p.t, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st)
    call(); torch.cuda.synchronize(); ref = out.clone()
    call(); call(); torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{dim}D {kind} n={n} kernel={p.kernel} cells={nc}: {dt*1e3:8.3f} ms  {nc/dt:10.1f} solves/s  bad={int((info != 0).sum())}  "
          f"bitwise equal to the first call: {bool(torch.equal(out, ref))}")
