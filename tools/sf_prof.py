"""Dev tool: per-phase clock counts (100 MHz wall clock) of k_small_fused for cell 0, from a library built with -DHOMMX_SF_PROF:
    hipcc -DHOMMX_SF_PROF ... -c hommx_amd/csrc/small.hip -o tools/bin/small_prof.o ; link it instead of small.o into tools/bin/lib_sfprof.so
    HOMMX_LIB=$PWD/tools/bin/lib_sfprof.so python tools/sf_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hommx_amd import MicroCellPlan, _lib

lib = _lib.load()
names = ["pre-invert", "invert", "mfma x2", "load rows", "sparse"]
for dim, n, kind in ((2, 10, "elasticity"), (3, 6, "poisson"), (3, 8, "poisson")):
    p = MicroCellPlan(dim, n, kind)
    nc = 2048
    shape = (nc, p.n_el) + ((p.n_comp,) if p.n_comp > 1 else ())
    coef = np.random.default_rng(0).uniform(0.5, 2.5, size=shape)
    buf = (ctypes.c_longlong * 16)()
    lib.hommx_sf_prof_read(buf, 1)
    p.solve(coef)
    lib.hommx_sf_prof_read(buf, 1)
    tot = sum(buf[:5])
    print(f"{dim}D {kind} n={n}: total {tot} ticks (10 ns each) = {tot/100:.1f} us per cell;", {k: buf[i] for i, k in enumerate(names)})
