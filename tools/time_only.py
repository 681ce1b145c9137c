"""Dev tool: timing only (results may be wrong under ablation builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hommx_amd import MicroCellPlan
dev = torch.device("cuda:0")
n, nc = 32, 16384
p = MicroCellPlan(2, n, "poisson")
coef = torch.rand(nc, 2 * n * n, dtype=torch.float64, device=dev) * 4.95 + 0.05
out = torch.empty(nc, 2, 2, dtype=torch.float64, device=dev)
info = torch.zeros(nc, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{ms:.3f} ms  {nc/ms*1e3:.3e} solves/s")
