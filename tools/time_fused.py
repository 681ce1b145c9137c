"""Dev tool: parity + timing of the fused 2D kernel on the GPU box (python tools/time_fused.py [n] [ncells])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan, workloads as W
from oracle import hommx_oracle as O

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ok = True
for n in (3, 5, 15, 16, 17, 31, 32):
    p = MicroCellPlan(2, n, "poisson")
    coef = rng.uniform(0.05, 5.0, size=(4, 2 * n * n))
    M = np.eye(2)[None] + 0.4 * rng.standard_normal((4, 2, 2))
    for MM in (None, M):
        A, info = p.solve(coef, MM, return_info=True)
        ref = O.effective_tensor_batch("poisson", 2, n, coef, MM)
        err = np.abs(A - ref).max() / np.abs(ref).max()
        ok &= bool(err < 1e-10) and not info.any()
print("parity", "OK" if ok else "FAIL")
for n, nc in ((16, 32768), (32, 8192), (32, 32768)):
    p = MicroCellPlan(2, n, "poisson")
    coef = torch.rand(nc, 2 * n * n, dtype=torch.float64, device=dev) * 4.95 + 0.05
    out = torch.empty(nc, 2, 2, dtype=torch.float64, device=dev)
    info = torch.zeros(nc, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 10
    e0.record()
    for _ in range(K):
        p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print(f"n={n} nc={nc}: {ms:.3f} ms/batch  {nc/ms*1e3:.3e} solves/s")
