"""Dev/benchmark tool: the blocked path on a C4-shaped batch (3D elasticity, 16^3 micro cells, fibre Hooke tensor)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan, workloads as W

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
msh, coef, _ = W.c4_fibre_beam(shape=(max(1, nc // 36), 1, 1) if nc >= 36 else (1, 1, 1), n=n)
coef = coef[:nc] if coef.shape[0] >= nc else np.tile(coef, (-(-nc // coef.shape[0]), 1, 1))[:nc]
p = MicroCellPlan(3, n, "elasticity")
dev = torch.device("cuda:0")
dc = torch.from_numpy(np.ascontiguousarray(coef)).to(dev)
out = torch.empty(nc, 6, 6, dtype=torch.float64, device=dev)
info = torch.zeros(nc, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
p.solve_device(nc, dc.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
t0 = time.time()
p.solve_device(nc, dc.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
dt = time.time() - t0
b = 3 * n * n
print(f"n={n} b={b} cells={nc}: {dt*1e3:.1f} ms, {nc/dt:.1f} solves/s, model {(6*(n-1)+2)*b**3*nc/dt/1e12:.2f} TFLOP/s, info!=0: {int((info!=0).sum())}")
