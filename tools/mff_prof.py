"""Dev tool: per-phase clocks of the register-resident front kernel (library whose mf_front_bs3.hip was compiled with -DMFF_PROF [-DMFF_DEV_BS3], HOMMX_LIB points at it):
phases 0 tables, 1 build, 2 panel -> LDS + barrier, 3 sweep, 4 Y' + barrier, 5 updates, 6 barrier, 7 store (summed over wave 0 of every workgroup)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hommx_amd import MicroCellPlan, _lib
lib = _lib.load()
lib.hommx_mff_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 256
p = MicroCellPlan(3, 16, "elasticity")
p.reserve(nc)
dev = torch.device("cuda:0")
coef = (torch.rand((nc, p.n_el, 2), dtype=torch.float64) * 2 + 0.5).to(dev)
out = torch.empty(nc, 6, 6, dtype=torch.float64, device=dev); info = torch.zeros(nc, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
buf = np.zeros(16, np.uint64)
p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
lib.hommx_mff_prof_read(buf.ctypes.data, 1)
p.solve_device(nc, coef.data_ptr(), None, out.data_ptr(), info.data_ptr(), st); torch.cuda.synchronize()
lib.hommx_mff_prof_read(buf.ctypes.data, 1)
tot = buf[:8].sum()
names = ["tables", "build", "panel->LDS+barrier", "sweep", "Y'+barrier", "updates", "barrier", "store"]
for n, v in zip(names, buf[:8]):
    print(f"{n:20s} {int(v):16d} clocks  {100.0 * v / tot:5.1f} %")
