"""C3 only (PoissonStratifiedHMM, 128x128 macro, 32x32 micro, wavy laminate) on the device entry point: for rocprofv3 passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan, workloads as W

msh, coef, M = W.c3_wavy_laminate()
p = MicroCellPlan(2, 32, "poisson")
dev = torch.device("cuda:0")
nc = coef.shape[0]
dc, dM = torch.from_numpy(coef).to(dev), torch.from_numpy(M).to(dev)
out = torch.empty(nc, 2, 2, dtype=torch.float64, device=dev); info = torch.zeros(nc, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(2):
    p.solve_device(nc, dc.data_ptr(), dM.data_ptr(), out.data_ptr(), info.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    p.solve_device(nc, dc.data_ptr(), dM.data_ptr(), out.data_ptr(), info.data_ptr(), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
err = np.abs(out.cpu().numpy() - W.stratified_laminate_exact(M)).max()
print(f"C3: {nc} cells, {dt*1e3:.3f} ms per launch, {nc/dt:.4e} solves/s, max abs err vs closed form {err:.2e}, bad {int((info!=0).sum())}")
