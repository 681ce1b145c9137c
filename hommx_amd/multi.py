"""Several MI355X driven from ONE process through the C ABI alone (no torch): ``hommx_comm_*`` + ``hommx_solve_batch_multi``.

The macro cells are block-partitioned over the devices exactly as the reference partitions them over MPI ranks
(/root/reference/src/hommx/hmm.py:307-310); RCCL all-gathers the effective-tensor field (and the info flags) over xGMI.
``hommx_amd.dist`` is the one-process-per-GPU counterpart for callers that already run under torch.distributed.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .batch import MicroCellPlan


class MultiGpuSolver:
    """One plan per device + one RCCL communicator; ``solve`` has the signature of ``MicroCellPlan.solve`` (host arrays)."""

    def __init__(self, dim: int, n_micro: int, kind: str = "poisson", devices: list[int] | None = None):
        self._lib = _lib.load()
        if devices is None:
            devices = list(range(self._lib.hommx_device_count()))
        if not devices:
            raise _lib.HommxLibraryError("no HIP device visible: hommx_amd has no CPU fallback for the micro-cell solves")
        self.devices = list(devices)
        self.plans = [MicroCellPlan(dim, n_micro, kind, device=d) for d in self.devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _lib.check(self._lib.hommx_comm_init_all(C.byref(h), len(self.devices), arr), "hommx_comm_init_all")
        self._h = h
        p0 = self.plans[0]
        self.dim, self.n_el, self.n_comp, self.t = p0.dim, p0.n_el, p0.n_comp, p0.t

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hommx_comm_destroy(self._h)
            self._h = None
        for p in getattr(self, "plans", []):
            p.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def solve(self, coef: np.ndarray, M: np.ndarray | None = None, return_info: bool = False):
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        nc = coef.shape[0]
        if coef.size != nc * self.n_el * self.n_comp:
            raise ValueError(f"coef has shape {coef.shape}; expected ({nc}, {self.n_el}" + (f", {self.n_comp})" if self.n_comp > 1 else ")"))
        Mp = None
        if M is not None:
            M = np.ascontiguousarray(M, dtype=np.float64)
            if M.shape != (nc, self.dim, self.dim):
                raise ValueError(f"M has shape {M.shape}; expected ({nc}, {self.dim}, {self.dim})")
            Mp = M.ctypes.data
        out = np.empty((nc, self.t, self.t), dtype=np.float64)
        info = np.zeros(nc, dtype=np.int32)
        plans = (C.c_void_p * len(self.plans))(*[p._h for p in self.plans])
        if nc:
            _lib.check(self._lib.hommx_solve_batch_multi(self._h, plans, nc, coef.ctypes.data, Mp, out.ctypes.data, info.ctypes.data),
                       "hommx_solve_batch_multi")
        return (out, info) if return_info else out
